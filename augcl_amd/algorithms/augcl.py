"""AugCL (src/algorithms/augcl.py:9-89): a weak critic trained on shifted obs and a strong critic
trained on strongly augmented obs share one target, which tracks the WEAK critic; the actor
learns from the strong critic.  Entered through the curriculum hand-off `load_pretrained_agent`."""
from copy import deepcopy

from .. import augmentations, flat
from ..augmentations import random_shift  # noqa: F401  (imported by name in the reference, augcl.py:3)
from .rad import RAD
from .sac import SAC


class AugCL(RAD):
    def __init__(self, obs_shape, action_shape, args):
        super().__init__(obs_shape, action_shape, args)
        # as in the reference (augcl.py:13-14) the pre-hand-off weak critic is a detached copy whose
        # optimiser owns orphan tensors; it only becomes meaningful after load_pretrained_agent()
        self.critic_weak = deepcopy(self.critic)
        self._weak_group = flat.FlatGroup(flat.critic_order(self.critic_weak.named_parameters()))
        self.critic_weak_optimizer = flat.FlatAdam(flat.FlatGroup(flat.critic_order(deepcopy(self.critic).named_parameters())),
                                                   self._hp["critic_lr"], (self._hp["critic_beta"], 0.999),
                                                   reducer=self._reducer)

    def load_pretrained_agent(self, pretrained_agent: SAC):
        """augcl.py:16-27: copy actor / critic / target values, then ALIAS the previous agent's critic and
        its optimiser (state included) as the weak critic."""
        for src, dst in ((pretrained_agent.actor_group, self.actor_group),
                         (pretrained_agent.critic_group, self.critic_group),
                         (pretrained_agent.target_group, self.target_group)):
            assert src.names == dst.names
            dst.param.copy_(src.param)
        self.critic_weak = pretrained_agent.critic
        self._weak_group = pretrained_agent.critic_group
        self.critic_weak_optimizer = pretrained_agent.critic_optimizer
        own = self.critic.encoder.shared_cnn.precision  # the aliased critic now computes inside THIS agent's update()
        for m in self._cnn_modules():
            m.precision = own
        if self._act is not None:  # acting graphs captured earlier replay the old precision's kernels
            self._act["graphs"] = {}
        if self._graph is not None:  # graphs captured before the hand-off reference the old weak critic
            self.critic_weak_optimizer.make_capturable()
            self._graph.update(graphs={}, static=None, pool=None, seen=0)

    def update_critic(self, obs_aug, obs_shift, action, reward, next_obs, not_done, L=None, step=None):
        target_Q = self.calculate_target_Q(next_obs=next_obs, reward=reward, not_done=not_done)
        self._cuts(self.critic_weak)
        self._cuts(self.critic)
        q1, q2 = self.critic_weak(obs_shift, action)
        critic_loss_shift = self._q_loss(q1, q2, target_Q)
        q1, q2 = self.critic(obs_aug, action)
        critic_loss_aug = self._q_loss(q1, q2, target_Q)
        if L is not None:
            self._log(L, "train_critic_weak/loss", critic_loss_shift, step)
            self._log(L, "train_critic_strong/loss", critic_loss_aug, step)
        # zero_grad / backward / step of the weak critic (augcl.py:50-52), then of the strong one (augcl.py:54-56).  The
        # two networks share nothing, so both backward passes run first -- with several ranks the weak critic's gradient
        # exchange is then in flight during the strong critic's backward -- and the two Adam steps follow.
        weak = self.critic_weak_optimizer.backward_begin(critic_loss_shift, self._cuts(self.critic_weak, False))
        strong = self.critic_optimizer.backward_begin(critic_loss_aug, self._cuts(self.critic, False))
        self.critic_weak_optimizer.backward_finish(weak)
        self.critic_optimizer.backward_finish(strong)

    def soft_update_critic_target(self):
        self._soft_update_from(self._weak_group)  # the target tracks the weak critic (augcl.py:58-67)

    def _optimizers(self):
        return [self.critic_optimizer, self.actor_optimizer, self.critic_weak_optimizer]

    def _front_end(self, obs, action, reward, next_obs, not_done, out=None):
        """augcl.py:70-72"""
        obs_shift = augmentations.random_shift(obs)
        obs_aug = self.apply_aug(obs.clone() if self._aug_mutates() else obs)  # on the UN-shifted obs (augcl.py:72)
        return obs_shift, obs_aug, action, reward, next_obs, not_done

    def _learn(self, t, L, step, do_actor, do_target):
        obs_shift, obs_aug, action, reward, next_obs, not_done = t
        self.update_critic(obs_shift=obs_shift, obs_aug=obs_aug, action=action, reward=reward, next_obs=next_obs,
                           not_done=not_done, L=L, step=step)
        if do_actor:
            self.update_actor_and_alpha(obs_aug, L, step)
        if do_target:
            self.soft_update_critic_target()
