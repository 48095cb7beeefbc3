"""Deterministic parameters / replay batches / noise for parity work.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``repo_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it.

Everything here is derived from ``numpy.random.RandomState`` so that the golden
generator (which runs the reference in the build container), the oracle and the
HIP path all see bit-identical parameters, inputs and noise without having to
store them in fixtures ("Fixture recipe").

State-dict names and shapes follow the reference modules:
  encoder            /root/reference/algorithms/repo/models/encoder.py:21-41
  transition_model   /root/reference/algorithms/repo/models/rssm.py:8-32
  obs_model          /root/reference/algorithms/repo/models/decoder.py:28-39
  reward_model       /root/reference/algorithms/repo/models/decoder.py:178-187
  actor_model        /root/reference/algorithms/repo/models/actor_critic.py:50-74
  value_model        /root/reference/algorithms/repo/models/actor_critic.py:9-18
"""
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np

MODULES = (
    "encoder",
    "transition_model",
    "obs_model",
    "reward_model",
    "actor_model",
    "value_model",
)
# modules whose parameters form ``model_params`` (dreamer.py:90-95)
MODEL_MODULES = ("encoder", "transition_model", "obs_model", "reward_model")
# TIA (tia.py:27-82): obs_model becomes the 6-channel TIAObservationModel and these modules are added;
# model_params in the reference's order
TIA_EXTRA_MODULES = ("distractor_transition_model", "distractor_obs_model", "distractor_only_obs_model",
                     "distractor_reward_model", "mask_head")
TIA_MODEL_MODULES = ("encoder", "transition_model", "reward_model", "obs_model", "distractor_transition_model",
                     "distractor_reward_model", "distractor_obs_model", "distractor_only_obs_model", "mask_head")


def default_config(**over):
    """Hot-path keys of experiments/train_repo.py:8-76 with their defaults."""
    c = dict(
        algo="repo",
        pixel_obs=True,
        embedding_size=1024,
        hidden_size=200,
        belief_size=200,
        state_size=30,
        dense_activation_function="elu",
        cnn_activation_function="relu",
        batch_size=50,
        chunk_size=50,
        horizon=15,
        gamma=0.99,
        gae_lambda=0.95,
        action_noise=0.0,
        action_ent_coef=3e-4,
        latent_ent_coef=0.0,
        free_nats=3,
        model_lr=3e-4,
        actor_lr=8e-5,
        value_lr=8e-5,
        grad_clip_norm=100.0,
        target_kl=3.0,
        beta_lr=1e-4,
        init_beta=1e-5,
        prior_train_steps=5,
        disag_model=False,
        inv_dynamics=False,
        disag_coef=0.0,
        tia_obs_coef=1.0,
        tia_adv_coef=1.0,
        tia_reward_train_steps=1,
        replay_size=1000,
        train_steps=1,
        prefill=0,
        load_checkpoint=False,
        load_offline=False,
        save_buffer=False,
        num_steps=0,
        train_every=500,
        eval_every=5000,
        checkpoint_every=25000,
        log_every=500,
    )
    c.update(over)
    return SimpleNamespace(**c)


def param_shapes(action_size, belief=200, state=30, hidden=200, embed=1024, image=64, tia=False, cond=0):
    """Ordered {module: OrderedDict(name -> shape)} in state_dict order.
    image=128: the BUILD-DEFINED 128 x 128 stack (no reference model exists at that size: the reference's encoder
    hard-codes the 64 x 64 flatten, encoder.py:39): the same four encoder convs (-> 256x6x6) + fc 9216 -> embed; the
    decoder's conv4 becomes 32 -> 16 (k6, 30 -> 64) and a conv5 16 -> 3 (k2, 64 -> 128) follows.
    cond > 0: the task-conditioned modules of the multitask agents (models/encoder.py:68-88, models/decoder.py:96-123,
    198-213, models/rssm.py:187-210, models/actor_critic.py:28-55,104-139): FiLM layers behind the conv stacks, `cond`
    more input columns in fc_embed_state_action and every head's fc1 (the pixel decoder's fc1 stays belief + state wide:
    ConditionalVisualObservationModel concatenates nothing, decoder.py:116)."""
    A = action_size + cond      # pseudo-actions [action | condition]
    feat = belief + state + cond
    assert image in (64, 128), image
    assert not cond or (image == 64 and not tia)
    enc = OrderedDict()
    for i, (co, ci) in enumerate([(32, 3), (64, 32), (128, 64), (256, 128)], 1):
        enc[f"conv{i}.weight"] = (co, ci, 4, 4)
        enc[f"conv{i}.bias"] = (co,)
    if image == 128:
        enc["fc.weight"] = (embed, 256 * 6 * 6)
        enc["fc.bias"] = (embed,)
    if cond:
        enc["film.weight"] = (2 * (32 + 64 + 128 + 256), cond)
        enc["film.bias"] = (2 * (32 + 64 + 128 + 256),)
    rssm = OrderedDict(
        [
            ("fc_embed_state_action.weight", (belief, state + A)),
            ("fc_embed_state_action.bias", (belief,)),
            ("rnn.weight_ih", (3 * belief, belief)),
            ("rnn.weight_hh", (3 * belief, belief)),
            ("rnn.bias_ih", (3 * belief,)),
            ("rnn.bias_hh", (3 * belief,)),
            ("fc_embed_belief_prior.weight", (hidden, belief)),
            ("fc_embed_belief_prior.bias", (hidden,)),
            ("fc_state_prior.weight", (2 * state, hidden)),
            ("fc_state_prior.bias", (2 * state,)),
            ("fc_embed_belief_posterior.weight", (hidden, belief + embed)),
            ("fc_embed_belief_posterior.bias", (hidden,)),
            ("fc_state_posterior.weight", (2 * state, hidden)),
            ("fc_state_posterior.bias", (2 * state,)),
        ]
    )
    dec = OrderedDict(
        [
            ("fc1.weight", (embed, belief + state)),  # the PIXEL decoder's fc1 never sees the condition (FiLM only)
            ("fc1.bias", (embed,)),
            ("conv1.weight", (embed, 128, 5, 5)),
            ("conv1.bias", (128,)),
            ("conv2.weight", (128, 64, 5, 5)),
            ("conv2.bias", (64,)),
            ("conv3.weight", (64, 32, 6, 6)),
            ("conv3.bias", (32,)),
            ("conv4.weight", (32, 3, 6, 6)),
            ("conv4.bias", (3,)),
        ]
    )
    if image == 128:
        dec["conv4.weight"], dec["conv4.bias"] = (32, 16, 6, 6), (16,)
        dec["conv5.weight"], dec["conv5.bias"] = (16, 3, 2, 2), (3,)
    if cond:
        dec["film.weight"] = (2 * (128 + 64 + 32), cond)
        dec["film.bias"] = (2 * (128 + 64 + 32),)

    def mlp(n_hidden_layers, out):
        d = OrderedDict()
        d["fc1.weight"] = (hidden, feat)
        d["fc1.bias"] = (hidden,)
        for i in range(2, n_hidden_layers + 1):
            d[f"fc{i}.weight"] = (hidden, hidden)
            d[f"fc{i}.bias"] = (hidden,)
        d[f"fc{n_hidden_layers + 1}.weight"] = (out, hidden)
        d[f"fc{n_hidden_layers + 1}.bias"] = (out,)
        return d

    out = OrderedDict(
        [
            ("encoder", enc),
            ("transition_model", rssm),
            ("obs_model", dec),
            ("reward_model", mlp(3, 1)),
            ("actor_model", mlp(4, 2 * action_size)),
            ("value_model", mlp(3, 1)),
        ]
    )
    if tia:
        # TIAObservationModel (models/decoder.py:154-165): conv4 emits 6 channels = [recon | mask]
        assert image == 64
        dec6 = OrderedDict(dec)
        dec6["conv4.weight"], dec6["conv4.bias"] = (32, 6, 6, 6), (6,)
        out["obs_model"] = dec6
        out["distractor_transition_model"] = OrderedDict(rssm)
        out["distractor_obs_model"] = OrderedDict(dec6)
        out["distractor_only_obs_model"] = OrderedDict(dec)
        out["distractor_reward_model"] = mlp(3, 1)
        out["mask_head"] = OrderedDict([("0.weight", (1, 6, 1, 1)), ("0.bias", (1,))])
    return out


def make_params(action_size, seed=7, image=64, tia=False, cond=0):
    """{module: OrderedDict(name -> float32 ndarray)}; uniform(-k, k), k = fan_in**-0.5.

    One RandomState drawn sequentially in (module, state_dict) order; a bias uses
    the bound of the weight registered just before it (rnn.bias_* use weight_hh's).
    """
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for mod, shapes in param_shapes(action_size, image=image, tia=tia, cond=cond).items():
        d = OrderedDict()
        k = 1.0
        for name, shp in shapes.items():
            if len(shp) > 1:
                k = 1.0 / np.sqrt(float(np.prod(shp[1:])))
            d[name] = rs.uniform(-k, k, size=shp).astype(np.float32)
        out[mod] = d
    return out


def make_batch(L, B, action_size, seed=11, planted_dones=((3, 1), (5, 2)), p_done=0.0, image=64):
    """Synthetic replay batch, time-major like SequenceReplayBuffer.sample
    (/root/reference/common/buffers.py:156-166)."""
    rs = np.random.RandomState(seed)
    obs = rs.randint(0, 256, size=(L, B, 3, image, image)).astype(np.uint8)
    actions = rs.uniform(-1, 1, size=(L, B, action_size)).astype(np.float32)
    rewards = rs.uniform(0, 1, size=(L, B, 1)).astype(np.float32)
    dones = (rs.uniform(size=(L, B, 1)) < p_done).astype(np.float32)
    for t, b in planted_dones:
        if t < L and b < B:
            dones[t, b, 0] = 1.0
    return obs, actions, rewards, dones


def make_tasks(L, B, num_tasks, seed=11):
    """Task one-hots (L, B, num_tasks) float32 as MultitaskSequenceReplayBuffer.sample returns them first
    (/root/reference/common/buffers.py:222-225).  A task is constant within an episode, but a sampled window may
    straddle an episode boundary: every other sequence switches task part-way."""
    rs = np.random.RandomState(seed + 7919)
    first = rs.randint(0, num_tasks, size=B)
    second = (first + 1 + rs.randint(0, max(num_tasks - 1, 1), size=B)) % num_tasks
    switch = rs.randint(1, max(L, 2), size=B)
    idx = np.where((np.arange(L)[:, None] >= switch[None, :]) & (np.arange(B)[None, :] % 2 == 1), second[None, :],
                   first[None, :])
    return np.eye(num_tasks, dtype=np.float32)[idx]


def preprocess_u8(obs_u8):
    """u8 -> f32 in [-1, 1]; same expression as /root/reference/common/utils.py:79."""
    return ((obs_u8.astype(np.float32) / 255) * 2) - 1.0


def make_noise(L, B, H, action_size, state=30, samples=100, seed=101, tia=False):
    """Noise for ONE update in the reference's draw order:
    train_dynamics: for t: eps_prior (B,S), eps_post (B,S);
    train_actor_critic: for t<H-1: eps_act (N,A), eps_prior (N,S); then entropy (100,(H-1)N,A).
    tia=True: the distractor filter's scan draws (d_obs_prior, d_obs_post) after the task scan's (tia.py:88-121).
    """
    rs = np.random.RandomState(seed)
    T = L - 1
    N = T * B

    def n(*shape):
        return rs.standard_normal(shape).astype(np.float32)

    obs_prior, obs_post = [], []
    for _ in range(T):
        obs_prior.append(n(B, state))
        obs_post.append(n(B, state))
    d_prior, d_post = [], []
    if tia:
        for _ in range(T):
            d_prior.append(n(B, state))
            d_post.append(n(B, state))
    img_act, img_prior = [], []
    for _ in range(H - 1):
        img_act.append(n(N, action_size))
        img_prior.append(n(N, state))
    ent = n(samples, (H - 1) * N, action_size)
    out = dict(
        obs_prior=np.stack(obs_prior),
        obs_post=np.stack(obs_post),
        img_act=np.stack(img_act),
        img_prior=np.stack(img_prior),
        entropy=ent,
    )
    if tia:
        out["d_obs_prior"], out["d_obs_post"] = np.stack(d_prior), np.stack(d_post)
    return out
