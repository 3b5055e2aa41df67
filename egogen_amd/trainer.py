"""On-policy collection + training loop reproducing the schedule the reference gets from tianshou
(`Collector`, `VectorReplayBuffer`, `onpolicy_trainer`; crowd_ppo/main_ppo.py:177-243):

  per epoch:  while steps_in_epoch < step_per_epoch:
                  collect step_per_collect transitions (= step_per_collect / A vector steps)
                  process_fn (values + GAE)  ->  learn(batch_size, repeat_per_collect)
              test: episode_per_test episodes, deterministic if --deterministic-eval
              save_best_fn (policy.pth) on improvement, save_checkpoint_fn every save_interval epochs

Environments are a batched `VecCrowdEnv` per rank; with torch.distributed initialised every rank owns
A/world agents and gradients are all-reduced inside `GAMMAPPOPolicy.learn`.
"""
from __future__ import annotations

import json
import os
import time
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from .crowd_env import VecCrowdEnv
from .ppo_policy import GAMMAPPOPolicy, RolloutBatch


class Collector:
    """Steps a VecCrowdEnv with the policy and fills a RolloutBatch (time-major).  Episode statistics are
    accumulated on the device; nothing synchronises with the host during collection."""

    def __init__(self, policy: GAMMAPPOPolicy, env: VecCrowdEnv, rollout_dir: Optional[str] = None):
        self.policy, self.env = policy, env
        self.A = env.A
        dev = env.dev
        self.ep_ret = torch.zeros(self.A, device=dev)
        self.ep_len = torch.zeros(self.A, device=dev)
        self._done = torch.zeros(3, device=dev)  # sum of returns, sum of lengths, count of the finished episodes
        self.done_ret, self.done_len, self.done_cnt = self._done[0], self._done[1], self._done[2]
        self._pol_out: dict = {}
        self._batches: Dict[int, RolloutBatch] = {}
        self.obs = None
        self.rollout_dir = rollout_dir
        self._episodes = [[] for _ in range(self.A)] if rollout_dir else None

    def reset(self):
        self.obs = self.env.reset()
        self.ep_ret.zero_()
        self.ep_len.zero_()
        self.reset_stat()

    def reset_stat(self):
        self._done.zero_()

    def _track(self, rew, term):
        if rew.is_cuda and rew.dtype == torch.float32 and term.dtype == torch.int32:
            from . import _lib
            lib = _lib.load()
            _lib.check(lib.egx_track_episode(_lib.ptr(rew), _lib.ptr(term), self.A, _lib.ptr(self.ep_ret), _lib.ptr(self.ep_len),
                                             _lib.ptr(self._done), _lib.current_stream_ptr()), "egx_track_episode")
            return
        self.ep_ret += rew
        self.ep_len += 1
        d = term.to(self.ep_ret.dtype)
        self.done_ret += (self.ep_ret * d).sum()
        self.done_len += (self.ep_len * d).sum()
        self.done_cnt += d.sum()
        self.ep_ret *= (1 - d)
        self.ep_len *= (1 - d)

    def _fusable(self, obs, rew, term) -> bool:
        """egx_rollout_store reads raw pointers: fp32 state[A,2,402] / egosensing[A,2,32] / dist[A] / time[A] / reward[A] and
        int32 terminated[A], all contiguous on the device.  Anything else (an env wrapper returning bool flags, views, other
        dtypes) goes through the converting copy_ path instead."""
        A = self.A
        want = (("state", (A, 2, 402)), ("egosensing", (A, 2, 32)), ("dist", (A,)), ("time", (A,)))
        for k, shape in want:
            t = obs.get(k)
            if t is None or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shape:
                return False
        return (rew.is_cuda and rew.dtype == torch.float32 and rew.is_contiguous() and tuple(rew.shape) == (A,) and
                term.is_cuda and term.dtype == torch.int32 and term.is_contiguous() and tuple(term.shape) == (A,))

    def _save_rollouts(self, term):
        """save_rollout_results per finished episode (crowd_env_2f.py:154-155,305-309) - host side, only when asked."""
        from .utils import save_rollout_results
        env = self.env
        mb = env.marker_b.cpu()
        pp = env.pred_params.cpu()
        fr = env.prev_frame.cpu()
        pel = env.joints.reshape(self.A, 20, -1, 3)[:, :, 0].cpu()
        tm = term.cpu().numpy()
        for a in range(self.A):
            mp = [mb[a:a + 1], pp[a:a + 1], env.betas[a].cpu(), getattr(env, "gender", "male"), fr[a, :9].reshape(3, 3), fr[a, 9:].reshape(1, 3), pel[a:a + 1], "2-frame"]
            self._episodes[a].append(mp)
            if tm[a]:
                scene = {"wpath": self._wpath_before[a], "navmesh_path": "synthetic"}
                save_rollout_results(scene, self._episodes[a], self.rollout_dir)
                self._episodes[a] = []

    def collect(self, n_vec_steps: int) -> RolloutBatch:
        if self.obs is None:
            self.reset()
        b = self._batches.get(n_vec_steps)
        if b is None:
            b = RolloutBatch(n_vec_steps, self.A, self.env.dev)
            self._batches[n_vec_steps] = b
        fused = self._episodes is None and self._fusable(self.obs, self.env.reward, self.env.terminated)
        if fused:
            from . import _lib
            lib, ptr = _lib.load(), _lib.ptr
            store = lambda o, t1, rew, term, t0: _lib.check(lib.egx_rollout_store(
                ptr(o["state"]), ptr(o["egosensing"]), ptr(o["dist"]), ptr(o["time"]), ptr(rew) if rew is not None else None,
                ptr(term) if term is not None else None, self.A, ptr(b.state[t1]), ptr(b.ego[t1]), ptr(b.dist[t1]), ptr(b.time[t1]),
                ptr(b.rew[t0]) if rew is not None else None, ptr(b.term[t0]) if rew is not None else None,
                ptr(self.ep_ret) if rew is not None else None, ptr(self.ep_len), ptr(self._done), _lib.current_stream_ptr()),
                "egx_rollout_store")
            store(self.obs, 0, None, None, 0)
        for t in range(n_vec_steps):
            if not fused:
                b.store_obs(t, self.obs)
            # the policy writes straight into the rollout slot of this step
            out = self.policy(self.obs, out={"act": b.act[t], "mu": b.mu[t], "logvar": b.logvar[t], "logp": b.logp_old[t],
                                             "value": b.values[t]})
            if self._episodes is not None:
                self._wpath_before = self.env.wpath.cpu()
            if self._episodes is not None:
                obs, rew, term = self.env.step(out["act"], auto_reset=False)
                self._save_rollouts(term)
                self.env.reset(mask=term)
            else:
                obs, rew, term = self.env.step(out["act"])
            if fused:
                # one launch: next observation -> slot t+1, reward / termination -> slot t, episode bookkeeping
                store(obs, t + 1, rew, term, t)
            else:
                b.rew[t].copy_(rew)
                b.term[t].copy_(term)
                self._track(rew, term)
            self.obs = obs
        if not fused:
            b.store_obs(n_vec_steps, self.obs)
        b.rollout_values_valid = True
        return b

    def collect_episodes(self, n_episode: int, max_steps: int = 64) -> Dict[str, float]:
        """Test collector: run until every env has finished one episode (n_episode == number of test envs in
        main_ppo.py:239-243); returns mean reward / length of those first episodes."""
        self.reset()
        dev = self.env.dev
        first_ret = torch.zeros(self.A, device=dev)
        first_len = torch.zeros(self.A, device=dev)
        got = torch.zeros(self.A, device=dev)
        run_ret = torch.zeros(self.A, device=dev)
        run_len = torch.zeros(self.A, device=dev)
        for _ in range(max_steps):
            out = self.policy(self.obs, out=self._pol_out)
            if self._episodes is not None:
                self._wpath_before = self.env.wpath.cpu()
                obs, rew, term = self.env.step(out["act"], auto_reset=False)
                self._save_rollouts(term)
                self.env.reset(mask=term)
            else:
                obs, rew, term = self.env.step(out["act"])
            run_ret += rew
            run_len += 1
            d = term.to(run_ret.dtype) * (1 - got)
            first_ret += run_ret * d
            first_len += run_len * d
            got = torch.clamp(got + d, max=1)
            run_ret *= (1 - term.to(run_ret.dtype))
            run_len *= (1 - term.to(run_ret.dtype))
            self.obs = obs
            if bool((got.sum() >= min(n_episode, self.A)).item()):
                break
        n = max(float(got.sum().item()), 1.0)
        return {"rew": float(first_ret.sum().item()) / n, "len": float(first_len.sum().item()) / n, "n/ep": n}


class ScalarLogger:
    """TensorBoard SummaryWriter when importable (the reference logs with tianshou's TensorboardLogger,
    main_ppo.py:192-205), else newline-delimited JSON with the same scalar names."""

    def __init__(self, log_path: str):
        os.makedirs(log_path, exist_ok=True)
        self.path = log_path
        self.writer = None
        try:
            from torch.utils.tensorboard import SummaryWriter  # noqa
            self.writer = SummaryWriter(log_path)
        except Exception:
            self.f = open(os.path.join(log_path, "scalars.jsonl"), "a")

    def write(self, prefix: str, step: int, data: Dict[str, float]):
        for k, v in data.items():
            if self.writer is not None:
                self.writer.add_scalar(f"{prefix}/{k}", v, step)
            else:
                self.f.write(json.dumps({"tag": f"{prefix}/{k}", "step": step, "value": v}) + "\n")
        if self.writer is None:
            self.f.flush()


class CheckpointWriter:
    """`torch.save` off the training loop.  The reference's drivers write `checkpoint_<epoch>.pth` / `policy.pth` synchronously
    (main_ppo.py:186-216); at this loop's speed an epoch of the default 20 000 env-steps takes ~0.1 s and a 158 MB checkpoint (model +
    AdamW moments) ~0.3 s, so the writes would be most of the wall time.  `save(state, path)` copies the tensors of `state` to host
    memory on the calling thread (the only part that must see the parameters of THIS moment) and queues the file write for one worker
    thread: files appear in submission order, at most `depth` snapshots wait in memory, `close()` (also at interpreter exit) returns when
    everything is on disk.  The file is what `torch.save` of the same dict writes, with the tensors located on the CPU - the
    reference's `torch.load(path, map_location=device)` (main_ppo.py:137-141) reads either.  `EGX_SYNC_CHECKPOINTS=1` writes in place."""

    def __init__(self, depth: int = 2):
        import atexit
        import queue
        import threading
        self.sync = os.environ.get("EGX_SYNC_CHECKPOINTS", "0") == "1"
        self._q = queue.Queue(maxsize=max(1, depth))
        self._err = None
        self._thread = None
        # device tensors are staged through page-locked host buffers (one per snapshot that can be alive: `depth` queued + one
        # being written), each tensor a view of its snapshot's buffer: ~4 ms instead of ~20 for 96 MB
        self._stage = [None] * (max(1, depth) + 1)
        self._free = queue.Queue()
        for i in range(len(self._stage)):
            self._free.put(i)
        if not self.sync:
            self._thread = threading.Thread(target=self._run, name="egx-checkpoint-writer", daemon=True)
            self._thread.start()
            atexit.register(self.close)

    @staticmethod
    def _walk(obj, fn):
        if torch.is_tensor(obj):
            return fn(obj)
        if isinstance(obj, dict):
            return {k: CheckpointWriter._walk(v, fn) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(CheckpointWriter._walk(v, fn) for v in obj)
        return obj

    @staticmethod
    def _to_host(obj):
        """plain copies (no staging buffer): what `save` falls back to for states without device tensors"""
        return CheckpointWriter._walk(obj, lambda t: t.detach().to("cpu", copy=True) if t.device.type == "cpu" else t.detach().cpu())

    def _snapshot(self, state, slot: int):
        """Host copy of `state`: every device tensor becomes a view of the slot's page-locked buffer of its dtype (torch.save
        writes a storage once and refuses views of different types on one storage: one buffer per dtype)."""
        # entries that alias one tensor (the policy's state_dict lists the actor / critic parameters under `actor.*`, `critic.*`
        # AND `_actor_critic.*`) are copied once and stay aliases in the file, as in a plain torch.save of the device state
        key = lambda t: (t.data_ptr(), t.dtype, tuple(t.shape), tuple(t.stride()))
        need, seen = {}, set()

        def count(t):
            if t.is_cuda and key(t) not in seen:
                seen.add(key(t))
                need[t.dtype] = need.get(t.dtype, 0) + (t.numel() + 15) // 16 * 16
        self._walk(state, count)
        if not need:
            return self._to_host(state)
        bufs = self._stage[slot]
        if bufs is None:
            bufs = self._stage[slot] = {}
        for dt, n in need.items():
            # exactly the size this snapshot needs: torch.save writes a view's WHOLE storage, so a larger buffer kept from an
            # earlier, bigger state would be written out in full (checkpoints of one run all have the same size: no churn)
            if dt not in bufs or bufs[dt].numel() != n:
                bufs[dt] = torch.empty(n, dtype=dt, pin_memory=True)
        off = {dt: 0 for dt in need}
        done = {}

        def take(t):
            if not t.is_cuda:
                return t.detach().to("cpu", copy=True)
            k = key(t)
            if k in done:
                return done[k]
            n, o = t.numel(), off[t.dtype]
            view = bufs[t.dtype][o:o + n].reshape(t.shape)
            off[t.dtype] = o + (n + 15) // 16 * 16
            view.copy_(t.detach(), non_blocking=True)
            done[k] = view
            return view
        out = self._walk(state, take)
        torch.cuda.current_stream().synchronize()
        return out

    def _run(self):
        while True:
            job = self._q.get()
            try:
                if job is None:
                    return
                state, path, slot = job
                try:
                    tmp = path + ".tmp"
                    torch.save(state, tmp)
                    os.replace(tmp, path)          # a reader never sees a half-written file
                finally:
                    del state
                    self._free.put(slot)
            except BaseException as e:        # surfaced by the next save() / close()
                self._err = e
            finally:
                self._q.task_done()

    def _raise(self):
        if self._err is not None:
            e, self._err = self._err, None
            raise RuntimeError(f"checkpoint write failed: {e!r}") from e

    def save(self, state, path: str) -> str:
        if self.sync:
            tmp = path + ".tmp"
            torch.save(state, tmp)
            os.replace(tmp, path)              # a reader never sees a half-written file, in either mode
            return path
        self._raise()
        slot = self._free.get()            # waits while every staging buffer still belongs to an unwritten snapshot
        try:
            snap = self._snapshot(state, slot)
        except BaseException:
            self._free.put(slot)
            raise
        self._q.put((snap, path, slot))
        return path

    def flush(self):
        if not self.sync:
            self._q.join()
            self._raise()

    def close(self):
        if self.sync or self._thread is None:
            return
        self._q.join()
        self._q.put(None)
        self._thread.join()
        self._thread = None
        self._raise()


def onpolicy_trainer(policy: GAMMAPPOPolicy, train_collector: Collector, test_collector: Optional[Collector], max_epoch: int,
                     step_per_epoch: int, repeat_per_collect: int, episode_per_test: int, batch_size: int,
                     step_per_collect: int, save_best_fn: Optional[Callable] = None, logger: Optional[ScalarLogger] = None,
                     save_checkpoint_fn: Optional[Callable] = None, save_interval: int = 2, verbose: bool = True):
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    A_global = train_collector.A * world
    n_vec = max(1, step_per_collect // A_global)
    env_step, gradient_step = 0, 0
    best_reward, best_epoch = -float("inf"), 0
    t_start = time.time()
    # tianshou's BaseTrainer.reset [upstream] (the trainer main_ppo.py:221-235 calls): one evaluation BEFORE epoch 1 fixes
    # best_reward / best_epoch = 0, and save_best_fn is called once so that policy.pth exists from the start
    if test_collector is not None:
        policy.eval()
        r0 = test_collector.collect_episodes(episode_per_test)
        best_reward, best_epoch = r0["rew"], 0
        if logger is not None and rank == 0:
            logger.write("test", 0, {"reward": r0["rew"], "length": r0["len"]})
        if verbose and rank == 0:
            print(f"Epoch #0: test_reward: {r0['rew']:.6f}", flush=True)
    if save_best_fn is not None and rank == 0:
        save_best_fn(policy)
    train_collector.reset()
    forced_seen = 0
    for epoch in range(1, max_epoch + 1):
        policy.train()
        steps_in_epoch = 0
        while steps_in_epoch < step_per_epoch:
            batch = train_collector.collect(n_vec)
            # device-side NaN/Inf counter, polled once per collect; with several ranks the counter is MAX-reduced first so
            # that every rank raises together instead of one leaving the others inside the next all-reduce
            train_collector.env.check_finite(all_ranks=world > 1)
            n_new = n_vec * A_global
            env_step += n_new
            steps_in_epoch += n_new
            policy.process_fn(batch)
            losses = policy.learn(batch, batch_size, repeat_per_collect)
            gradient_step += len(losses["loss"])
            if logger is not None and rank == 0:
                cnt = float(train_collector.done_cnt.item())
                if cnt > 0:
                    logger.write("train", env_step, {"reward": float(train_collector.done_ret.item()) / cnt,
                                                     "length": float(train_collector.done_len.item()) / cnt, "n/ep": cnt})
                logger.write("update", env_step, {k: float(np.mean(v)) for k, v in losses.items() if v})
            train_collector.reset_stat()  # on every rank
        result = None
        if test_collector is not None:
            policy.eval()
            result = test_collector.collect_episodes(episode_per_test)
            if logger is not None and rank == 0:
                logger.write("test", env_step, {"reward": result["rew"], "length": result["len"]})
            if result["rew"] > best_reward:
                best_reward, best_epoch = result["rew"], epoch
                if save_best_fn is not None and rank == 0:
                    save_best_fn(policy)
        if save_checkpoint_fn is not None and rank == 0 and epoch % save_interval == 0:
            save_checkpoint_fn(epoch, env_step, gradient_step)
        # box scenes: an agent none of whose reset draws passed the start check starts in penetration (the reference's
        # `while True` loop, crowd_env_2f_box.py:349-416, would still be drawing): say so, once per epoch (one host sync)
        fa = getattr(train_collector.env, "forced_accepts", None)
        if fa is not None:
            n_forced = fa()
            if n_forced > forced_seen:
                import warnings
                warnings.warn(f"epoch {epoch}: {n_forced - forced_seen} episode(s) started without a valid start among their reset draws "
                              f"({n_forced} since construction)")
                if logger is not None and rank == 0:
                    logger.write("train", env_step, {"forced_accepts": float(n_forced)})
                forced_seen = n_forced
        if verbose and rank == 0:
            msg = f"Epoch #{epoch}: env_step {env_step}, gradient_step {gradient_step}"
            if result is not None:
                msg += f", test_reward: {result['rew']:.6f}, best_reward: {best_reward:.6f} in #{best_epoch}"
            print(msg, flush=True)
    dur = time.time() - t_start
    return {"duration": f"{dur:.2f}s", "train_step": env_step, "best_reward": best_reward, "best_epoch": best_epoch,
            "train_speed": f"{env_step / max(dur, 1e-9):.2f} step/s", "gradient_step": gradient_step}
