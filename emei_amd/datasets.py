"""Offline datasets from GPU rollouts, in the schema of the reference's data generator
(zoo/util.py:33-93 `rollout`, :16-30 `get_replay_buffer`, keys checked by emei/core.py:118-126):

    observations, next_observations, actions, rewards, dones, timeouts

The reference fills these with a Python loop over ONE env (`env.step` per sample).  Here one fused
rollout launch with device auto-reset produces N trajectories at once; `observations` after a reset
are rebuilt from the device reset generator (emei_episode_init_obs), so every row is a true
transition (obs, a, r, next_obs, done, timeout).  Rows are ordered env-major (each env's samples are
contiguous and in time order), i.e. like the reference's episode-after-episode concatenation.
"""
import json

import numpy as np
import torch

DATASET_KEYS = ("observations", "next_observations", "actions", "rewards", "dones", "timeouts")


def random_actions(env, n_steps, generator=None):
    """`env.action_space.sample()` for every (step, env): uniform ints / uniform box (zoo/util.py:58)."""
    eng = env.engine
    if eng.act_dim == 0:
        return torch.randint(0, env.action_space.n, (n_steps, eng.n_envs), device=eng.device, dtype=torch.uint8,
                             generator=generator)
    lo = torch.as_tensor(env.action_space.low, device=eng.device, dtype=torch.float32)
    hi = torch.as_tensor(env.action_space.high, device=eng.device, dtype=torch.float32)
    shape = (n_steps, eng.n_envs) if eng.act_dim == 1 else (n_steps, eng.n_envs, eng.act_dim)
    u = torch.rand(shape + (() if eng.act_dim > 1 else ()), device=eng.device, generator=generator)
    if eng.act_dim == 1:
        return (lo[0] + (hi[0] - lo[0]) * u).float().contiguous()
    return (lo + (hi - lo) * u).float().contiguous()


def collect(env, n_steps, actions=None, seed=0, device_rng=True, policy=None):
    """Roll every env of `env` (a vectorised emei_amd env) for n_steps with auto-reset and return
    (dataset dict of torch tensors with N*n_steps rows, rollout_info dict like zoo/util.py:85-91).

    Actions: `actions` [T, N(, act_dim)] if given; else `policy(obs [N, obs_dim] float32 on the device) ->
    actions [N(, act_dim)]` evaluated between steps (the agent.predict loop of zoo/util.py:54-59: one
    emei_step launch per step, observations after a device reset rebuilt the same way); else uniform random
    actions (`env.action_space.sample()`, zoo/util.py:58) in ONE fused rollout launch."""
    eng = env.engine
    N, T, od = eng.n_envs, int(n_steps), eng.obs_dim
    obs0, _ = env.reset(seed=seed, options={"device_rng": True} if device_rng else None)
    obs0 = torch.as_tensor(obs0, device=eng.device).reshape(N, od).float()
    _, epi0 = eng.get_counters()
    if actions is None and policy is not None:
        next_obs, rew, done = eng.alloc_outputs(T)
        acts_t, cur, epi = [], obs0, epi0.clone()
        for t in range(T):
            a = policy(cur)
            a = a.to(eng.device)
            a = (a.to(torch.float32) if eng.act_dim else a.to(torch.int64)).contiguous()
            acts_t.append(a)
            eng.step(a, auto_reset=True, out=(next_obs[t], rew[t], done[t]))
            cur = next_obs[t]
            d = done[t] != 0
            if bool(d.any()):  # the policy must see what reset() returned for the finished envs
                epi = epi + d.to(torch.int64)
                idx = torch.nonzero(d).reshape(-1)
                cur = cur.clone()
                cur[idx] = eng.episode_init_obs(idx, epi[idx])
        actions = torch.stack(acts_t)
    else:
        if actions is None:
            g = torch.Generator(device=eng.device)
            g.manual_seed(int(seed))
            actions = random_actions(env, T, g)
        next_obs, rew, done = eng.rollout(actions, auto_reset=True)
    done_b = done != 0
    # observations[t] = next_observations[t-1], except right after a reset
    obs = torch.empty_like(next_obs)
    obs[0] = obs0
    obs[1:] = next_obs[:-1]
    prev_done = torch.zeros_like(done_b)
    prev_done[1:] = done_b[:-1]
    if bool(prev_done.any()):
        t_idx, e_idx = torch.nonzero(prev_done, as_tuple=True)
        episodes = epi0[None, :] + torch.cumsum(done_b.to(torch.int64), dim=0)  # episode index AFTER step t's reset
        obs[t_idx, e_idx] = eng.episode_init_obs(e_idx, episodes[t_idx - 1, e_idx])
    em = lambda x: x.transpose(0, 1).reshape((N * T,) + tuple(x.shape[2:]))  # env-major rows
    acts = actions if actions.dim() == 3 else actions[..., None]
    data = {
        "observations": em(obs),
        "next_observations": em(next_obs),
        "actions": em(acts.to(torch.float32) if eng.act_dim else acts),
        "rewards": em(rew),
        "dones": em(done_b.float()),               # float(terminated or truncated), zoo/util.py:66
        "timeouts": em(((done & 2) != 0).float()),  # float(truncated), zoo/util.py:67
    }
    n_epi = int(done_b.sum())
    info = dict(avg_reward=float(rew.sum() / max(n_epi, 1)), avg_length=float(N * T / max(n_epi, 1)),
                total_episode_num=n_epi, total_sample_num=N * T)
    return data, info


def _host_arrays(dataset):
    return {k: v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v) for k, v in dataset.items()}


def _write_info(path, rollout_info):
    if rollout_info is not None:
        with open(str(path) + ".info.json", "w") as f:
            json.dump(rollout_info, f, indent=4)


def save_h5(dataset, path, rollout_info=None):
    """`save_as_h5` of the reference (zoo/util.py:108-111): one HDF5 dataset per key in the root group of a new file, in
    the on-disk structures h5py itself uses for that call (emei_amd/h5io.py; no h5py needed), so that the reference's
    `load_h5_data` (emei/core.py:61-81) — or any h5py / libhdf5 reader — opens it."""
    from . import h5io

    h5io.write_h5(path, _host_arrays(dataset))
    _write_info(path, rollout_info)


def load_h5(path):
    """`OfflineEnv.load_h5_data` (emei/core.py:61-81) + the key check of :118-126."""
    from . import h5io

    d = h5io.read_h5(path)
    for key in DATASET_KEYS:
        assert key in d, "Dataset is missing key %s" % key
    return d


def save_npz(dataset, path, rollout_info=None):
    """The same keys as a compressed .npz (this package's own container; the reference reads .h5: save_h5)."""
    np.savez_compressed(path, **_host_arrays(dataset))
    _write_info(path, rollout_info)


def save_for_env(env, dataset, dataset_name, rollout_info=None, fmt="h5"):
    """Write `dataset` where `env.get_dataset(dataset_name)` looks for it.
    fmt "h5" (default): `<root>/<env_name>-v0/<env_params_name>/<dataset_name>.h5` — exactly the path the REFERENCE's
    `get_dataset` resolves its URL to (core.py:82-91 with the URL scheme of offline_info.py:33-39) and loads from when the
    file already exists (core.py:95-103 downloads only a missing file), in the reference's container (zoo/util.py:108-111).
    fmt "npz": `<root>/<env_name>/<env_params_name>/<dataset_name>.npz`, rounds 1-3's layout."""
    if fmt not in ("h5", "npz"):
        raise ValueError(f"fmt {fmt!r}: 'h5' or 'npz'")
    d = env.reference_dataset_dir if fmt == "h5" else env.dataset_dir
    d.mkdir(parents=True, exist_ok=True)
    path = d / f"{dataset_name}.{fmt}"
    (save_h5 if fmt == "h5" else save_npz)(dataset, path, rollout_info)
    return path


def load_npz(path):
    d = dict(np.load(path))
    for key in DATASET_KEYS:  # emei/core.py:118-126
        assert key in d, "Dataset is missing key %s" % key
    return d
