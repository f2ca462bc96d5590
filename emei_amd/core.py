"""Host-side API surface of emei's core classes, re-implemented for the HIP engine.

Reference behaviour kept (file:line under /root/reference/emei/core.py):
  * ``Freezable`` (:18-37): a ``frozen`` flag with assertions on double freeze / unfreeze.
  * ``OfflineEnv`` (:40-58): ``env_name`` = class name without the "Env" suffix, ``env_params_name`` =
    "&"-joined ``key=value`` pairs in sorted key order (the string test/test_core.py pins).
    The download plumbing (:60-107) is a network fetch and is never performed; datasets are looked up
    in the reference's directory scheme (``<root>/<env_name>-v0/<env_params_name>/<dataset>.h5``, :82-91) as
    the HDF5 files ``emei_amd.datasets`` writes from GPU rollouts (``emei_amd/h5io.py``: the reference's own
    container, zoo/util.py:108-111, without h5py) or this package's ``.npz``, with the key check of :118-126.
  * ``EmeiEnv`` (:131-193): causal-graph getters and the abstract batched functions.
"""
import os
import pathlib

import numpy as np

# core.py:15 of the reference; EMEI_DATASET_PATH relocates it (tests, shared filesystems)
DATASET_PATH = pathlib.Path(os.environ.get("EMEI_DATASET_PATH", str(pathlib.Path.home() / ".emei" / "offline_data")))


class Freezable:
    """Snapshot switch used by model-based callers: freeze() -> query -> unfreeze()."""

    frozen_state = None
    frozen = False

    def __init__(self):
        self.frozen_state, self.frozen = None, False

    def _flip(self, to, complaint):
        assert self.frozen != to, complaint
        self.frozen = to

    def freeze(self):
        self._flip(True, "env has frozen")

    def unfreeze(self):
        self._flip(False, "env has unfrozen")


class OfflineEnv:
    """Naming metadata that keys the reference's offline datasets."""

    metadata = {}

    def __init__(self, env_params):
        cls = type(self).__name__
        self.env_name = cls[:-3]  # "CartPoleSwingUpEnv" -> "CartPoleSwingUp"
        self.env_params = env_params
        self._offline_dataset_names, self._offline_dataset_urls = [], {}

    @property
    def dataset_dir(self) -> pathlib.Path:
        """<root>/<env_name>/<env_params_name>: this package's .npz files (rounds 1-3)."""
        return DATASET_PATH / self.env_name / self.env_params_name

    @property
    def reference_dataset_dir(self) -> pathlib.Path:
        """<root>/<env_name>-v0/<env_params_name>: where the reference's get_path_from_url (core.py:82-91) puts the file of
        URL `.../<env_name>-v0/<params>/<dataset>.h5` (offline_info.py:33-39) — and where it finds one already there."""
        return DATASET_PATH / f"{self.env_name}-v0" / self.env_params_name

    def _dataset_files(self):
        """{dataset name: path}; an .h5 in the reference's directory wins over an .npz of the same name"""
        found = {}
        for d, pat in ((self.dataset_dir, "*.npz"), (self.reference_dataset_dir, "*.h5")):
            if d.is_dir():
                found.update({p.stem: p for p in sorted(d.glob(pat))})
        return found

    @property
    def dataset_names(self):
        """Datasets available for this (env, params): here the local files, never a URL table."""
        return sorted(self._dataset_files())

    @property
    def env_params_name(self):
        params = self.env_params
        return "&".join(f"{k}={params[k]}" for k in sorted(params))

    def get_dataset(self, dataset_name):
        """core.py:109-128 without the download: load the local `<dataset_name>.h5` (the reference's container, read by
        emei_amd/h5io.py as load_h5_data does, :61-81) or `.npz`, then run the same key check."""
        files = self._dataset_files()
        assert dataset_name in files, (
            f"dataset {dataset_name!r} not found under {self.reference_dataset_dir} (.h5) or {self.dataset_dir} (.npz): this "
            "build never downloads (core.py:95-107); generate one on the GPU with emei_amd.datasets.collect() + save_for_env()")
        path = files[dataset_name]
        if path.suffix == ".h5":
            from . import h5io

            data = h5io.read_h5(path)
        else:
            data = dict(np.load(path))
        for key in ["observations", "observations", "actions", "rewards", "dones", "timeouts"]:  # sic, core.py:118-126
            assert key in data, "Dataset is missing key %s" % key
        return data

    @staticmethod
    def load_h5_data(h5path):
        """core.py:61-81: {dataset path in the file: array} of every dataset of an HDF5 file"""
        from . import h5io

        return h5io.read_h5(h5path)


class EmeiEnv(Freezable, OfflineEnv):
    """Base of every env: graphs + the batched reward / terminal / init / next-obs interface."""

    def __init__(self, env_params):
        Freezable.__init__(self)
        OfflineEnv.__init__(self, env_params=env_params)
        self._transition_graph = self._reward_mech_graph = self._termination_mech_graph = None

    # -- causal metadata ---------------------------------------------------------------------------
    def get_transition_graph(self, repeat_times=1):
        """0/1 matrix [(n_obs + n_act), n_obs]: entry (i, j) says variable i influences obs j within
        `repeat_times` steps.  One step is the stored graph; more steps are walks of length <= k in the
        graph extended by zero columns for the action rows (core.py:142-161), computed here as boolean
        reachability instead of summed float matrix powers."""
        graph = self._transition_graph.copy()  # raises on None exactly like the reference (CartPole has no graph)
        n_obs, n_act = self.observation_space.shape[0], self.action_space.shape[0]
        assert graph.shape == (n_obs + n_act, n_obs)
        if repeat_times == 1:
            return graph
        step = np.zeros((n_obs + n_act, n_obs + n_act), dtype=bool)
        step[:, :n_obs] = graph.astype(bool)
        reach, walk = np.zeros_like(step), step.copy()
        for _ in range(repeat_times):
            reach |= walk
            walk = (walk.astype(np.int64) @ step.astype(np.int64)) > 0
        return reach[:, :n_obs].astype(int)

    def get_reward_mech_graph(self):
        return self._reward_mech_graph

    def get_termination_mech_graph(self):
        return self._termination_mech_graph

    # -- state <-> observation (identity unless an env overrides) -----------------------------------
    def transform_state_to_obs(self, batch_state):
        return batch_state.copy()

    def transform_obs_to_state(self, batch_obs):
        return batch_obs.copy()

    # -- batched functions every concrete env provides ---------------------------------------------
    def get_batch_init_state(self, batch_size):
        raise NotImplementedError

    def get_batch_init_obs(self, batch_size):
        return self.transform_state_to_obs(self.get_batch_init_state(batch_size=batch_size))

    def get_batch_reward(self, obs, pre_obs=None, action=None, state=None, pre_state=None):
        raise NotImplementedError

    def get_batch_terminal(self, obs, pre_obs=None, action=None, state=None, pre_state=None):
        raise NotImplementedError

    def get_batch_next_obs(self, obs, pre_obs=None, action=None, state=None, pre_state=None):
        assert self.frozen
        raise NotImplementedError
