"""Host-side mirror of emei's core API (reference: emei/core.py).

Same names, argument meaning and error behaviour as the reference's ``Freezable`` / ``OfflineEnv`` /
``EmeiEnv`` for the env-step path; the dataset download plumbing (core.py:60-128) is out of scope
(network fetch) and raises.
"""
from abc import abstractmethod
from typing import Dict, Union

import numpy as np


class Freezable:
    """core.py:18-37."""

    def __init__(self):
        self.frozen_state = None
        self.frozen = False

    def freeze(self):
        assert not self.frozen, "env has frozen"
        self.frozen = True

    def unfreeze(self):
        assert self.frozen, "env has unfrozen"
        self.frozen = False


class OfflineEnv:
    """core.py:40-128 without the network plumbing."""

    metadata = {}

    def __init__(self, env_params: Dict[str, Union[str, int, float]]):
        self.env_name = self.__class__.__name__[:-3]  # strip "Env" (core.py:42)
        self.env_params = env_params
        self._offline_dataset_urls = {}
        self._offline_dataset_names = []

    @property
    def dataset_names(self) -> list:
        return self._offline_dataset_names

    @property
    def env_params_name(self):
        """core.py:56-58: '&'.join('k=v' for sorted k)."""
        return "&".join("{}={}".format(key, self.env_params[key]) for key in sorted(self.env_params.keys()))

    def get_dataset(self, dataset_name: str):
        raise NotImplementedError(
            "offline datasets are fetched over the network by the reference (core.py:95-128); "
            "use emei_amd.datasets.collect() to generate them on the GPU instead"
        )


class EmeiEnv(Freezable, OfflineEnv):
    """core.py:131-193."""

    def __init__(self, env_params: Dict[str, Union[str, int, float]]):
        Freezable.__init__(self)
        OfflineEnv.__init__(self, env_params=env_params)
        self._transition_graph = None
        self._reward_mech_graph = None
        self._termination_mech_graph = None

    def get_transition_graph(self, repeat_times=1):
        """core.py:142-161: (n_obs+n_act) x n_obs 0/1 graph; repeat_times > 1 = reachability within
        that many steps through repeated products of the square-augmented graph."""
        g = self._transition_graph.copy()  # AttributeError on None, like the reference (CartPole defines no graph)
        num_obs, num_action = self.observation_space.shape[0], self.action_space.shape[0]
        assert g.shape == (num_obs + num_action, num_obs)
        if repeat_times == 1:
            return g
        aug_g = np.zeros([num_obs + num_action, num_obs + num_action])
        aug_g[:, :num_obs] = g.copy()
        prod_g = aug_g.copy()
        sum_g = np.zeros([num_obs + num_action, num_obs + num_action])
        for _ in range(repeat_times):
            sum_g += prod_g
            prod_g = np.matmul(prod_g, aug_g)
        return (sum_g > 0).astype(int)[:, :num_obs]

    def get_reward_mech_graph(self):
        return self._reward_mech_graph

    def get_termination_mech_graph(self):
        return self._termination_mech_graph

    def transform_state_to_obs(self, batch_state):
        return batch_state.copy()

    def transform_obs_to_state(self, batch_obs):
        return batch_obs.copy()

    @abstractmethod
    def get_batch_init_state(self, batch_size):
        raise NotImplementedError

    def get_batch_init_obs(self, batch_size):
        return self.transform_state_to_obs(self.get_batch_init_state(batch_size=batch_size))

    @abstractmethod
    def get_batch_reward(self, obs, pre_obs=None, action=None, state=None, pre_state=None):
        raise NotImplementedError

    @abstractmethod
    def get_batch_terminal(self, obs, pre_obs=None, action=None, state=None, pre_state=None):
        raise NotImplementedError

    @abstractmethod
    def get_batch_next_obs(self, obs, pre_obs=None, action=None, state=None, pre_state=None):
        assert self.frozen
        raise NotImplementedError
