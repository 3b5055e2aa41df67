import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tests.helpers import load_golden, rebuild_state_dict, max_abs
from egogen_amd.models import *
g = load_golden("policy_ref.npz")
sd = rebuild_state_dict(g, g["fill_seeds"], ["shared_net.", "actor.", "critic."], gains=[1.0, 1.4, 1.4])
ac = ActorCritic(GAMMAActor(POLICY_CFG), GAMMACritic(POLICY_CFG), GAMMAPolicyBase(POLICY_CFG))
ac.load_state_dict(sd, strict=True); ac.cuda()
obs = {k[4:]: torch.from_numpy(v).cuda() for k, v in g.items() if k.startswith("obs_")}
hx = ac.shared_net(obs)
ref = g["hx"]
print("manual-gru hx parts", [max_abs(hx[:, a:b].detach().cpu(), ref[:, a:b]) for a, b in ((0,512),(512,1024),(1024,1088),(1088,1152))])
_, h2 = ac.shared_net.x_enc(obs["state"].permute(1,0,2))
print("nn.GRU(MIOpen) x_enc", max_abs(h2[0].detach().cpu(), ref[:, :512]))
(mu, lv), _ = ac.actor(hx)
print("mu", max_abs(mu.detach().cpu(), g["mu"]))
