"""Network forwards of the hot path as explicit tensor math over reference-keyed state dicts.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates (reference file:line):
  cvae_decode        models/models_GAMMA_primitive.py:83-101,113-133  (GAMMAPrimitiveVAE.decode / sample_prior)
  regressor_6d       models/models_GAMMA_primitive.py:160-175,222-259 (ResNetBlock, MoshRegressor._forward)
  regressor_forward  models/models_GAMMA_primitive.py:208-219,262-301 (+ _cont2aa, baseops.py:119-162)
  sample_prior       models/models_GAMMA_primitive.py:334-360         (GAMMAPrimitiveCombo.sample_prior)
  policy_base / actor / critic   models/models_policy_ppo.py:24-39,276-306,326-330,348-350; baseops.py:615-641
  vposer_encode      human_body_prior 1.0 VPoser.encode (.loc)  [upstream, PARITY UNPINNED], call site crowd_env_2f.py:198
Pinned by tests/golden/{cvae,regressor,policy}_ref.npz generated from the importable reference classes.
"""
import torch
import torch.nn.functional as F

from .rot import cont2aa


def linear(x, sd, prefix):
    return x @ sd[prefix + ".weight"].t() + sd[prefix + ".bias"]


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRU / GRUCell equations, gate order (r, z, n)."""
    H = h.shape[-1]
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def gru_last_hidden(x_seq, sd, prefix):
    """nn.GRU(in, H) over x_seq[t,b,in], zero initial state -> last hidden [b,H]."""
    w_ih, w_hh = sd[prefix + ".weight_ih_l0"], sd[prefix + ".weight_hh_l0"]
    b_ih, b_hh = sd[prefix + ".bias_ih_l0"], sd[prefix + ".bias_hh_l0"]
    h = torch.zeros(x_seq.shape[1], w_hh.shape[1], dtype=x_seq.dtype)
    for t in range(x_seq.shape[0]):
        h = gru_cell(x_seq[t], h, w_ih, w_hh, b_ih, b_hh)
    return h


def mlp(x, sd, prefix, n_layers, act):
    """baseops.py:638-641: activation after EVERY linear."""
    for i in range(n_layers):
        x = act(linear(x, sd, f"{prefix}.layers.{i}"))
    return x


def cvae_decode(sd, x, z, t_pred=None, prefix="predictor."):
    """x[t_his,b,201], z[b,128] -> y[t_pred,b,201]; residual, use_drnn_mlp (cfg MPVAE_samp20_2frame_rollout.yml)."""
    if t_pred is None:
        t_pred = 20 - x.shape[0]
    hx = gru_last_hidden(x, sd, prefix + "x_enc")
    h_rnn = mlp(hx, sd, prefix + "drnn_mlp", 3, torch.tanh)
    ys = []
    y_i = None
    for i in range(t_pred):
        y_p = x[-1][:, :201] if i == 0 else y_i
        rnn_in = torch.cat([hx, z, y_p], dim=-1)
        h_rnn = gru_cell(rnn_in, h_rnn, sd[prefix + "d_rnn.weight_ih"], sd[prefix + "d_rnn.weight_hh"],
                         sd[prefix + "d_rnn.bias_ih"], sd[prefix + "d_rnn.bias_hh"])
        hfc = mlp(h_rnn, sd, prefix + "d_mlp", 2, torch.tanh)
        y_i = linear(hfc, sd, prefix + "d_out") + y_p
        ys.append(y_i)
    return torch.stack(ys)


def resnet_block(x, sd, prefix, n_blocks=10):
    h = linear(x, sd, prefix + ".in_fc")
    for b in range(n_blocks):
        h = mlp(h, sd, f"{prefix}.layers.{b}", 2, torch.relu) + h
    return linear(h, sd, prefix + ".out_fc")


def regressor_6d(sd, markers, betas, n_recur=3, prefix="regressor."):
    """markers[n,201], betas[n,10] -> xb6d[n,159] (transl3 | 22x6D | hands 24), zero init, 3 recurrences."""
    n = markers.shape[0]
    xb = torch.zeros(n, 159, dtype=markers.dtype)
    for _ in range(n_recur):
        xb = resnet_block(torch.cat([markers, xb, betas], dim=-1), sd, prefix + "pnet") + xb
    return xb


def regressor_forward(sd, markers, betas, prefix="regressor."):
    """-> xb[n,93] with axis-angle rotations (use_cont: true)."""
    xb = regressor_6d(sd, markers, betas, prefix=prefix)
    n = xb.shape[0]
    aa = cont2aa(xb[:, 3:3 + 132].contiguous().view(n, -1, 6)).reshape(n, -1)
    return torch.cat([xb[:, :3], aa[:, :3], aa[:, 3:], xb[:, 135:147], xb[:, 147:]], dim=-1)


def sample_prior(sd, X, betas, z):
    """X[2,b,201], betas[18,b,10], z[b,128] -> Y[18,b,201], Yb[18,b,93]."""
    Y = cvae_decode(sd, X, z)
    nt, nb = Y.shape[:2]
    Yb = regressor_forward(sd, Y.reshape(nt * nb, -1), betas.reshape(nt * nb, -1)).view(nt, nb, -1)
    return Y, Yb


# ---- policy (models_policy_ppo.py) ------------------------------------------------------------

def positional_encoding(x, L=32):
    """models_policy_ppo.py:276-285: [sin(x*2^k), cos(x*2^k)] interleaved per k; x[b,1] -> [b,2L]."""
    outs = []
    for k in range(L):
        f = 2.0 ** k
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


def policy_base(sd, obs, prefix="shared_net."):
    """obs{state[b,2,402], egosensing[b,2,32], dist[b,1] or [b], time[b,1] or [b]} -> hx[b,1152]."""
    nb = obs["state"].shape[0]
    hx = gru_last_hidden(obs["state"].permute(1, 0, 2), sd, prefix + "x_enc")
    he = gru_last_hidden(obs["egosensing"].permute(1, 0, 2), sd, prefix + "ego_enc")
    d = positional_encoding(obs["dist"].reshape(nb, 1))
    t = positional_encoding(obs["time"].reshape(nb, 1))
    return torch.cat([hx, he, d, t], dim=-1)


def mlp_block(h, sd, prefix, n_blocks=2):
    lrelu = lambda v: F.leaky_relu(v, 0.01)
    for b in range(n_blocks):
        h = mlp(h, sd, f"{prefix}.layers.{b}", 2, lrelu) + h
    return linear(h, sd, prefix + ".out_fc")


def policy_actor(sd, hx, prefix="actor."):
    zp = mlp_block(hx, sd, prefix + "pnet")
    return zp[:, :128], zp[:, 128:]


def policy_critic(sd, hx, prefix="critic."):
    return mlp_block(hx, sd, prefix + "vnet")


# ---- VPoser v1 encoder ----------------------------------------------------------------------

def vposer_encode(sd, body_pose):
    """human_body_prior 1.0 VPoser.encode(...).loc in eval mode: BN(63) -> fc1(63,512) lrelu(.2) ->
    BN(512) -> [dropout off] -> fc2(512,512) lrelu(.2) -> mu(512,32).  body_pose[n,63] -> [n,32]."""
    def bn(x, p):
        return (x - sd[p + ".running_mean"]) / torch.sqrt(sd[p + ".running_var"] + 1e-5) * sd[p + ".weight"] + sd[p + ".bias"]
    x = bn(body_pose, "bodyprior_enc_bn1")
    x = F.leaky_relu(linear(x, sd, "bodyprior_enc_fc1"), 0.2)
    x = bn(x, "bodyprior_enc_bn2")
    x = F.leaky_relu(linear(x, sd, "bodyprior_enc_fc2"), 0.2)
    return linear(x, sd, "bodyprior_enc_mu")
