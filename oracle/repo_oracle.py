"""CPU oracle: a from-scratch PyTorch (fp32, CPU) restatement of RePo's update.

TEST INFRASTRUCTURE ONLY -- the product package ``repo_amd`` never imports this
module.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  The reference holds no tests or golden vectors for this
path; the oracle is instead pinned against outputs of the
reference itself, produced in the build container by
``tests/golden/gen_golden.py`` (committed together with its .npz outputs) and
checked by ``tests/test_oracle_golden.py``.

What it restates (file:line in /root/reference):
  encoder                 algorithms/repo/models/encoder.py:34-41
  observe / cell          algorithms/repo/models/rssm.py:34-64,76-146
  imagine                 algorithms/repo/models/rssm.py:148-184
  decoder, reward head    algorithms/repo/models/decoder.py:41-48,189-195
  actor, value            algorithms/repo/models/actor_critic.py:20-26,76-102
  tanh-normal, entropy    algorithms/repo/models/utils.py:112-163
  lambda_return           common/utils.py:61-71
  RePo.train_dynamics     algorithms/repo/repo.py:25-112
  Dreamer.train_dynamics  algorithms/repo/dreamer.py:241-302
  train_actor_critic      algorithms/repo/dreamer.py:304-381
  clip + Adam             torch.nn.utils.clip_grad_norm_ / torch.optim.Adam as called at
                          repo.py:86-96, dreamer.py:356-359,370-373 (torch==1.12.1 semantics)

Differences from the reference are deliberate and limited to plumbing: all noise
is an explicit input (same draw order), parameters live in flat dicts keyed by
the reference's state_dict names, and the optimiser is written out.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import fixtures as fx

LOG_2PI = math.log(2.0 * math.pi)
ATANH_CLAMP = 0.99999997  # literal of models/utils.py:128 (rounds to 0.99999994 in fp32)


# --------------------------------------------------------------------------- modules
# Test hook (None = plain relu, the reference's arithmetic): callable (conv index 1..4, pre-activation, relu(pre)) -> the
# layer's output.  A ReLU whose pre-activation lies within fp32 rounding of zero has no decision two correct fp32
# convolutions must agree on, and the pixel's whole gradient rides on it; a gradient-parity test hands the oracle the
# decisions of the implementation under test INSIDE that band (tests/test_tia_gpu.py) and nowhere else.
RELU_TIE_BREAK = None


def encoder_fwd(p, obs):
    """obs (rows,3,64,64) f32 -> (rows,1024).  encoder.py:34-41 (fc is Identity at embedding_size 1024).
    128 x 128 frames (build-defined stack, fixtures.param_shapes(image=128)): the flatten is 9216 wide and
    `fc` = Linear(9216, 1024), applied without an activation like the reference's optional fc (encoder.py:40)."""
    h = obs
    for i in range(1, 5):
        pre = F.conv2d(h, p[f"conv{i}.weight"], p[f"conv{i}.bias"], stride=2)
        h = F.relu(pre)
        if RELU_TIE_BREAK is not None:
            h = RELU_TIE_BREAK(i, pre, h)
    h = h.reshape(h.shape[0], -1)
    if "fc.weight" in p:
        h = F.linear(h, p["fc.weight"], p["fc.bias"])
    return h


def gru_cell(p, x, h):
    """nn.GRUCell semantics, gates ordered r,z,n (rssm.py:24,39)."""
    gi = F.linear(x, p["rnn.weight_ih"], p["rnn.bias_ih"])
    gh = F.linear(h, p["rnn.weight_hh"], p["rnn.bias_hh"])
    H = h.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H : 2 * H] + gh[:, H : 2 * H])
    n = torch.tanh(gi[:, 2 * H :] + r * gh[:, 2 * H :])
    return (1 - z) * n + z * h


def compute_belief(p, prev_belief, state, action):
    hid = F.elu(
        F.linear(torch.cat([state, action], 1), p["fc_embed_state_action.weight"], p["fc_embed_state_action.bias"])
    )
    return gru_cell(p, hid, prev_belief)


def gaussian_head(p, prefix_embed, prefix_state, x, eps, min_std=0.1):
    hid = F.elu(F.linear(x, p[prefix_embed + ".weight"], p[prefix_embed + ".bias"]))
    out = F.linear(hid, p[prefix_state + ".weight"], p[prefix_state + ".bias"])
    S = out.shape[1] // 2
    mean, raw = out[:, :S], out[:, S:]
    std = F.softplus(raw) + min_std
    return mean + std * eps, mean, std


def observe(p, prev_belief, prev_state, actions, embeds, nonterms, eps_prior, eps_post):
    """rssm.py:76-146 with observations and nonterminals given.  Returns 7 stacked tensors."""
    T = actions.shape[0]
    outs = [[] for _ in range(7)]
    belief, post = prev_belief, prev_state
    for t in range(T):
        state = post * nonterms[t]
        belief = compute_belief(p, belief, state, actions[t])
        prior, pm, ps = gaussian_head(p, "fc_embed_belief_prior", "fc_state_prior", belief, eps_prior[t])
        post, qm, qs = gaussian_head(
            p, "fc_embed_belief_posterior", "fc_state_posterior", torch.cat([belief, embeds[t]], 1), eps_post[t]
        )
        for lst, v in zip(outs, (belief, prior, pm, ps, post, qm, qs)):
            lst.append(v)
    return [torch.stack(o, 0) for o in outs]


def decoder_fwd(p, belief, state):
    """decoder.py:41-48."""
    h = F.linear(torch.cat([belief, state], 1), p["fc1.weight"], p["fc1.bias"])
    h = h.view(-1, p["fc1.weight"].shape[0], 1, 1)
    h = F.relu(F.conv_transpose2d(h, p["conv1.weight"], p["conv1.bias"], stride=2))
    h = F.relu(F.conv_transpose2d(h, p["conv2.weight"], p["conv2.bias"], stride=2))
    h = F.relu(F.conv_transpose2d(h, p["conv3.weight"], p["conv3.bias"], stride=2))
    if "conv5.weight" in p:  # build-defined 128 x 128 stack: one more ReLU layer, then the output layer
        h = F.relu(F.conv_transpose2d(h, p["conv4.weight"], p["conv4.bias"], stride=2))
        return F.conv_transpose2d(h, p["conv5.weight"], p["conv5.bias"], stride=2)
    return F.conv_transpose2d(h, p["conv4.weight"], p["conv4.bias"], stride=2)


def mlp_head(p, belief, state, n_layers):
    """ELU MLP on cat([belief, state]); last layer linear.  decoder.py:189-195, actor_critic.py:20-26."""
    h = torch.cat([belief, state], 1)
    for i in range(1, n_layers):
        h = F.elu(F.linear(h, p[f"fc{i}.weight"], p[f"fc{i}.bias"]))
    return F.linear(h, p[f"fc{n_layers}.weight"], p[f"fc{n_layers}.bias"])


def scalar_head(p, belief, state):
    return mlp_head(p, belief, state, 4).squeeze(1)


def actor_fwd(p, belief, state, min_std=0.1, init_std=0.0, mean_scale=5.0):
    """actor_critic.py:76-87."""
    out = mlp_head(p, belief, state, 5)
    A = out.shape[1] // 2
    mean = mean_scale * torch.tanh(out[:, :A] / mean_scale)
    std = F.softplus(out[:, A:] + init_std) + min_std
    return mean, std


def tanh_normal_log_prob(y, mean, std):
    """log-prob of y under tanh(Normal(mean,std)), summed over the action dim.
    models/utils.py:126-134 (inverse recomputed from y, clamped) + torch's
    TransformedDistribution.log_prob / Normal.log_prob."""
    yc = torch.where(y.abs() <= 1.0, torch.clamp(y, -ATANH_CLAMP, ATANH_CLAMP), y)
    x = torch.atanh(yc)
    ladj = 2.0 * (math.log(2.0) - x - F.softplus(-2.0 * x))
    base = -((x - mean) ** 2) / (2 * std**2) - std.log() - 0.5 * LOG_2PI
    return (base - ladj).sum(-1)


def tanh_normal_entropy(mean, std, eps):
    """SampleDist.entropy (models/utils.py:160-163): eps is (samples, rows, A)."""
    y = torch.tanh(mean + std * eps)
    return -tanh_normal_log_prob(y, mean, std).mean(0)


def imagine(rssm, actor, belief0, state0, horizon, eps_act, eps_prior):
    """rssm.py:148-184 with policy = actor.get_action (rsample, detached inputs)."""
    beliefs, states, means, stds = [], [], [], []
    belief, state = belief0, state0
    for t in range(horizon - 1):
        a_mean, a_std = actor_fwd(actor, belief.detach(), state.detach())
        action = torch.tanh(a_mean + a_std * eps_act[t])
        belief = compute_belief(rssm, belief, state, action)
        state, pm, ps = gaussian_head(rssm, "fc_embed_belief_prior", "fc_state_prior", belief, eps_prior[t])
        beliefs.append(belief)
        states.append(state)
        means.append(pm)
        stds.append(ps)
    return [torch.stack(x, 0) for x in (beliefs, states, means, stds)]


def lambda_return(rewards, values, discounts, bootstrap, lambda_):
    """common/utils.py:61-71."""
    next_values = torch.cat([values[1:], bootstrap[None]], 0)
    inputs = rewards + discounts * next_values * (1 - lambda_)
    last = bootstrap
    outs = []
    for t in reversed(range(inputs.shape[0])):
        last = inputs[t] + discounts[t] * lambda_ * last
        outs.append(last)
    return torch.stack(outs[::-1], 0)


def normal_kl(qm, qs, pm, ps):
    """KL(N(qm,qs) || N(pm,ps)) elementwise (torch.distributions.kl._kl_normal_normal)."""
    var_ratio = (qs / ps) ** 2
    t1 = ((qm - pm) / ps) ** 2
    return 0.5 * (var_ratio + t1 - 1 - var_ratio.log())


# --------------------------------------------------------------------------- optimiser
class Adam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0, amsgrad=False)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self):
        self.t += 1
        bc1 = 1 - self.b1**self.t
        bc2 = 1 - self.b2**self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-(self.lr / bc1))

    # torch >= 2.0: zero_grad() sets gradients to None (set_to_none=True); torch 1.12.1 -- the version the reference
    # pins (requirements.txt:17) -- ZEROES them, so a parameter whose gradient is not recomputed still takes an Adam
    # step on a zero gradient (moments decay, the parameter moves, the step count advances).  Only TIA's update
    # depends on the difference (OracleTIA, cfg.zero_grad_set_to_none).
    set_to_none = True

    def zero_grad(self):
        for p in self.params:
            if self.set_to_none or p.grad is None:
                p.grad = None
            else:
                p.grad = torch.zeros_like(p.grad)


@torch.no_grad()
def clip_grad_norm(params, max_norm):
    """torch.nn.utils.clip_grad_norm_ (L2): returns the pre-clip total norm."""
    grads = [p.grad for p in params if p.grad is not None]
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


# --------------------------------------------------------------------------- agent
class OracleAgent:
    """Holds parameters + optimiser state and runs reference-faithful updates on CPU."""

    def __init__(self, cfg, action_size, params=None, seed=7, image=64):
        self.c = cfg
        self.A = action_size
        np_params = params if params is not None else fx.make_params(action_size, seed, image=image)
        self.p = OrderedDict()
        for mod in fx.MODULES:
            self.p[mod] = OrderedDict(
                (k, torch.tensor(np.asarray(v), dtype=torch.float32).requires_grad_(True)) for k, v in np_params[mod].items()
            )
        self.model_params = [t for mod in fx.MODEL_MODULES for t in self.p[mod].values()]
        self.actor_params = list(self.p["actor_model"].values())
        self.value_params = list(self.p["value_model"].values())
        self.model_opt = Adam(self.model_params, cfg.model_lr)
        self.actor_opt = Adam(self.actor_params, cfg.actor_lr)
        self.value_opt = Adam(self.value_params, cfg.value_lr)
        self.is_repo = cfg.algo == "repo"
        if self.is_repo:
            self.log_beta = torch.tensor(np.log(cfg.init_beta), dtype=torch.float32, requires_grad=True)
            self.beta_opt = Adam([self.log_beta], cfg.beta_lr)
        self.last = {}

    # -- world model ---------------------------------------------------------------
    def train_dynamics(self, obs, actions, rewards, nonterms, eps_prior, eps_post, apply=True):
        c, p = self.c, self.p
        L, B = obs.shape[:2]
        T = L - 1
        embeds = encoder_fwd(p["encoder"], obs.reshape(L * B, *obs.shape[2:])).reshape(L, B, -1)
        b0 = torch.zeros(B, c.belief_size)
        s0 = torch.zeros(B, c.state_size)
        beliefs, prior_s, pm, ps, post_s, qm, qs = observe(
            p["transition_model"], b0, s0, actions[:-1], embeds[1:], nonterms[:-1], eps_prior, eps_post
        )
        fb, fs = beliefs.reshape(T * B, -1), post_s.reshape(T * B, -1)
        if self.is_repo:
            recon = decoder_fwd(p["obs_model"], fb.detach(), fs.detach())
        else:
            recon = decoder_fwd(p["obs_model"], fb, fs)
        recon = recon.reshape(T, B, *obs.shape[2:])
        obs_loss = (0.5 * (recon - obs[1:]) ** 2 + 0.5 * LOG_2PI).sum((2, 3, 4)).mean((0, 1))

        r_pred = scalar_head(p["reward_model"], fb, fs).reshape(T, B)
        r_tgt = rewards[:-1].squeeze(-1)
        mask = nonterms[:-1].squeeze(-1)
        reward_loss = ((0.5 * (r_pred - r_tgt) ** 2 + 0.5 * LOG_2PI) * mask).mean((0, 1))

        out = {}
        if self.is_repo:
            kl_prior = normal_kl(qm.detach(), qs.detach(), pm, ps).sum(2).mean((0, 1))
            kl_post = normal_kl(qm, qs, pm.detach(), ps.detach()).sum(2).mean((0, 1))
            alpha = c.prior_train_steps / (1 + c.prior_train_steps)
            kl_div = alpha * kl_prior + (1 - alpha) * kl_post
            kl_viol = kl_div - c.target_kl
            kl_loss = self.log_beta.exp().detach() * kl_viol
            out["train/kl_div"] = kl_div
        else:
            kl = normal_kl(qm, qs, pm, ps).sum(2)
            kl_loss = torch.max(kl, torch.full((1,), float(c.free_nats))).mean((0, 1))
        model_loss = obs_loss + reward_loss + kl_loss

        self.model_opt.zero_grad()
        model_loss.backward()
        self.last["model_grads"] = [None if q.grad is None else q.grad.detach().clone() for q in self.model_params]
        total = clip_grad_norm(self.model_params, c.grad_clip_norm)
        self.last["model_total_norm"] = float(total)
        if apply:
            self.model_opt.step()

        out.update(
            {
                "train/obs_loss": obs_loss,
                "train/reward_loss": reward_loss,
                "train/kl_loss": kl_loss,
                "train/model_loss": model_loss,
            }
        )
        if self.is_repo:
            beta_loss = -self.log_beta * kl_viol.detach()
            self.beta_opt.zero_grad()
            beta_loss.backward()
            if apply:
                self.beta_opt.step()
            out["train/beta"] = self.log_beta.exp()
            out["train/beta_loss"] = beta_loss
        self.last["embeds"] = embeds.detach()
        self.last["observe"] = [x.detach() for x in (beliefs, prior_s, pm, ps, post_s, qm, qs)]
        scal = {k: float(v.detach()) for k, v in out.items()}
        return beliefs.detach(), post_s.detach(), scal

    # -- actor critic --------------------------------------------------------------
    def train_actor_critic(self, beliefs, post_states, eps_act, eps_prior, eps_ent, apply=True):
        c, p = self.c, self.p
        H = c.horizon
        frozen = self.model_params + self.value_params
        for q in frozen:
            q.requires_grad_(False)
        try:
            ib, istate, im, isd = imagine(
                p["transition_model"], p["actor_model"], beliefs, post_states, H, eps_act, eps_prior
            )
            Hm, N = ib.shape[:2]
            fb, fs = ib.reshape(Hm * N, -1), istate.reshape(Hm * N, -1)
            r_pred = scalar_head(p["reward_model"], fb, fs).reshape(Hm, N)
            v_pred = scalar_head(p["value_model"], fb, fs).reshape(Hm, N)
        finally:
            for q in frozen:
                q.requires_grad_(True)
        a_mean, a_std = actor_fwd(p["actor_model"], fb, fs)
        action_entropy = tanh_normal_entropy(a_mean, a_std, eps_ent).mean()
        latent_entropy = (0.5 + 0.5 * LOG_2PI + isd.log()).sum(-1).mean()
        disc = c.gamma * torch.ones_like(r_pred)
        returns = lambda_return(r_pred[:-1], v_pred[:-1], disc[:-1], v_pred[-1], c.gae_lambda)
        actor_loss = -returns.mean() - c.action_ent_coef * action_entropy - c.latent_ent_coef * latent_entropy

        self.actor_opt.zero_grad()
        if getattr(self.model_opt, "set_to_none", True):   # (torch 1.12.1 semantics keep the stale gradients: they are
            for q in self.model_params + self.value_params:  # what the next zero_grad() zeroes instead of dropping)
                q.grad = None
        actor_loss.backward()
        self.last["actor_grads"] = [q.grad.detach().clone() for q in self.actor_params]
        total_a = clip_grad_norm(self.actor_params, c.grad_clip_norm)
        self.last["actor_total_norm"] = float(total_a)
        if apply:
            self.actor_opt.step()

        vb, vs = ib[:-1].detach().reshape((Hm - 1) * N, -1), istate[:-1].detach().reshape((Hm - 1) * N, -1)
        tgt = returns.detach().reshape(-1)
        v = scalar_head(p["value_model"], vb, vs)
        value_loss = (0.5 * (v - tgt) ** 2 + 0.5 * LOG_2PI).mean()
        self.value_opt.zero_grad()
        value_loss.backward()
        self.last["value_grads"] = [q.grad.detach().clone() for q in self.value_params]
        total_v = clip_grad_norm(self.value_params, c.grad_clip_norm)
        self.last["value_total_norm"] = float(total_v)
        if apply:
            self.value_opt.step()
        self.last["imagine"] = [x.detach() for x in (ib, istate, im, isd)]
        self.last["returns"] = returns.detach()
        return {
            "train/actor_loss": float(actor_loss.detach()),
            "train/value_loss": float(value_loss.detach()),
            "train/action_entropy": float(action_entropy.detach()),
            "train/latent_entropy": float(latent_entropy.detach()),
        }

    # -- one full update on a uint8 replay batch -----------------------------------------
    def update(self, obs_u8, actions, rewards, dones, noise):
        obs = torch.from_numpy(fx.preprocess_u8(np.asarray(obs_u8)))
        acts = torch.as_tensor(actions)
        rews = torch.as_tensor(rewards)
        nonterms = 1 - torch.as_tensor(dones)
        t = {k: torch.as_tensor(v) for k, v in noise.items()}
        beliefs, post, scal = self.train_dynamics(obs, acts, rews, nonterms, t["obs_prior"], t["obs_post"])
        scal.update(
            self.train_actor_critic(
                beliefs.flatten(0, 1), post.flatten(0, 1), t["img_act"], t["img_prior"], t["entropy"]
            )
        )
        return beliefs, post, scal

    def module_grad_norms(self):
        """Pre-clip... no: post-clip grads are stored in-place; use self.last[*_grads] (pre-clip copies)."""
        out, i = {}, 0
        for mod in fx.MODEL_MODULES:
            n = len(self.p[mod])
            sq = sum(float((g.double() ** 2).sum()) for g in self.last["model_grads"][i : i + n] if g is not None)
            out[mod] = math.sqrt(sq)
            i += n
        out["actor_model"] = math.sqrt(sum(float((g.double() ** 2).sum()) for g in self.last["actor_grads"]))
        out["value_model"] = math.sqrt(sum(float((g.double() ** 2).sum()) for g in self.last["value_grads"]))
        return out


# --------------------------------------------------------------------------- TIA
class OracleTIA(OracleAgent):
    """TIA(Dreamer), /root/reference/algorithms/repo/tia.py:17-209: a distractor RSSM on the same embeddings, a
    mask-blended pair of 6-channel decoders (models/decoder.py:154-175) + mask_head (tia.py:69), a distractor-only
    decoder, an adversarial (frozen) distractor reward head that is then fitted for tia_reward_train_steps steps.
    ONE Adam over model_params in the reference's order (tia.py:71-82): parameters whose gradient is None are
    skipped by a step (per-parameter step counts, as torch.optim.Adam keeps them).
    Parity status: PINNED by tests/golden/tia_tiny.npz (the reference's TIA run by tests/golden/gen_golden.py)."""

    def __init__(self, cfg, action_size, params=None, seed=7):
        np_params = params if params is not None else fx.make_params(action_size, seed, tia=True)
        super().__init__(cfg, action_size, params=np_params)
        assert not self.is_repo
        for mod in fx.TIA_EXTRA_MODULES:
            self.p[mod] = OrderedDict(
                (k, torch.tensor(np.asarray(v), dtype=torch.float32).requires_grad_(True)) for k, v in np_params[mod].items()
            )
        self.model_params = [t for mod in fx.TIA_MODEL_MODULES for t in self.p[mod].values()]
        self.model_opt = PerParamAdam(self.model_params, cfg.model_lr)
        # the reference pins torch==1.12.1 (zero_grad() zeroes); the goldens tia_tiny / tia_coefs were generated under
        # torch 2.x (sets to None), tia_zeros.npz with zero_grad patched to the 1.12.1 behaviour
        self.model_opt.set_to_none = bool(getattr(cfg, "zero_grad_set_to_none", True))

    def train_dynamics(self, obs, actions, rewards, nonterms, eps_prior, eps_post, d_eps_prior=None, d_eps_post=None,
                       apply=True):
        c, p = self.c, self.p
        L, B = obs.shape[:2]
        T = L - 1
        embeds = encoder_fwd(p["encoder"], obs.reshape(L * B, *obs.shape[2:])).reshape(L, B, -1)
        b0 = torch.zeros(B, c.belief_size)
        s0 = torch.zeros(B, c.state_size)
        tb, _, tpm, tps, tpost, tqm, tqs = observe(p["transition_model"], b0, s0, actions[:-1], embeds[1:], nonterms[:-1],
                                                   eps_prior, eps_post)
        db, _, dpm, dps, dpost, dqm, dqs = observe(p["distractor_transition_model"], b0, s0, actions[:-1], embeds[1:],
                                                   nonterms[:-1], d_eps_prior, d_eps_post)
        tfb, tfs = tb.reshape(T * B, -1), tpost.reshape(T * B, -1)
        dfb, dfs = db.reshape(T * B, -1), dpost.reshape(T * B, -1)
        # tia.py:123-133
        t_recon, t_mask = decoder_fwd(p["obs_model"], tfb, tfs).chunk(2, 1)
        d_recon, d_mask = decoder_fwd(p["distractor_obs_model"], dfb, dfs).chunk(2, 1)
        m = torch.sigmoid(F.conv2d(torch.cat((t_mask, d_mask), 1), p["mask_head"]["0.weight"], p["mask_head"]["0.bias"]))
        recon = (t_recon * m + d_recon * (1 - m)).reshape(T, B, *obs.shape[2:])
        obs_loss = (0.5 * (recon - obs[1:]) ** 2 + 0.5 * LOG_2PI).sum((2, 3, 4)).mean((0, 1))
        # tia.py:135-145
        d_only = decoder_fwd(p["distractor_only_obs_model"], dfb, dfs).reshape(T, B, *obs.shape[2:])
        d_obs_loss = (0.5 * (d_only - obs[1:]) ** 2 + 0.5 * LOG_2PI).sum((2, 3, 4)).mean((0, 1))
        # tia.py:147-158
        r_tgt = rewards[:-1].squeeze(-1)
        mask = nonterms[:-1].squeeze(-1)
        t_reward = scalar_head(p["reward_model"], tfb, tfs).reshape(T, B)
        frozen = list(p["distractor_reward_model"].values())
        for q in frozen:
            q.requires_grad_(False)
        try:
            d_reward = scalar_head(p["distractor_reward_model"], dfb, dfs).reshape(T, B)
        finally:
            for q in frozen:
                q.requires_grad_(True)
        t_reward_loss = ((0.5 * (t_reward - r_tgt) ** 2 + 0.5 * LOG_2PI) * mask).mean((0, 1))
        d_reward_loss = (-(0.5 * (d_reward - r_tgt) ** 2 + 0.5 * LOG_2PI) * mask).mean((0, 1))
        reward_loss = t_reward_loss + c.tia_adv_coef * d_reward_loss
        # tia.py:160-172
        free = torch.full((1,), float(c.free_nats))
        t_kl = normal_kl(tqm, tqs, tpm, tps).sum(2)
        d_kl = normal_kl(dqm, dqs, dpm, dps).sum(2)
        kl_loss = torch.max(t_kl, free).mean((0, 1)) + torch.max(d_kl, free).mean((0, 1))
        model_loss = obs_loss + c.tia_obs_coef * d_obs_loss + reward_loss + kl_loss
        self.model_opt.zero_grad()
        model_loss.backward()
        self.last["model_grads"] = [None if q.grad is None else q.grad.detach().clone() for q in self.model_params]
        total = clip_grad_norm(self.model_params, c.grad_clip_norm)
        self.last["model_total_norm"] = float(total)
        if apply:
            self.model_opt.step()
        # tia.py:184-196
        self.last["d_reward_total_norms"] = []
        for _ in range(int(c.tia_reward_train_steps)):
            d_reward = scalar_head(p["distractor_reward_model"], dfb.detach(), dfs.detach()).reshape(T, B)
            d_reward_loss = ((0.5 * (d_reward - r_tgt) ** 2 + 0.5 * LOG_2PI) * mask).mean((0, 1))
            self.model_opt.zero_grad()
            d_reward_loss.backward()
            self.last["d_reward_grads"] = [q.grad.detach().clone() for q in frozen]
            self.last["d_reward_total_norms"].append(float(clip_grad_norm(self.model_params, c.grad_clip_norm)))
            if apply:
                self.model_opt.step()
        out = {
            "train/obs_loss": obs_loss, "train/d_obs_loss": d_obs_loss, "train/reward_loss": reward_loss,
            "train/t_reward_loss": t_reward_loss, "train/d_reward_loss": d_reward_loss, "train/kl_loss": kl_loss,
            "train/t_kl_div": t_kl.mean(), "train/d_kl_div": d_kl.mean(), "train/model_loss": model_loss,
        }
        return tb.detach(), tpost.detach(), {k: float(v.detach()) for k, v in out.items()}

    def update(self, obs_u8, actions, rewards, dones, noise):
        obs = torch.from_numpy(fx.preprocess_u8(np.asarray(obs_u8)))
        t = {k: torch.as_tensor(v) for k, v in noise.items()}
        beliefs, post, scal = self.train_dynamics(obs, torch.as_tensor(actions), torch.as_tensor(rewards),
                                                  1 - torch.as_tensor(dones), t["obs_prior"], t["obs_post"],
                                                  t["d_obs_prior"], t["d_obs_post"])
        scal.update(self.train_actor_critic(beliefs.flatten(0, 1), post.flatten(0, 1), t["img_act"], t["img_prior"],
                                            t["entropy"]))
        return beliefs, post, scal


class PerParamAdam(Adam):
    """torch.optim.Adam's per-parameter step count: a parameter whose gradient is None is skipped entirely."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr, betas, eps)
        self.steps = [0] * len(self.params)

    @torch.no_grad()
    def step(self):
        for i, (p, m, v) in enumerate(zip(self.params, self.m, self.v)):
            if p.grad is None:
                continue
            self.steps[i] += 1
            t = self.steps[i]
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(1 - self.b2**t)).add_(self.eps)
            p.addcdiv_(m, denom, value=-(self.lr / (1 - self.b1**t)))


# --------------------------------------------------------------------------- encoder fine-tuning (adaptation)
class OracleFinetuned(OracleAgent):
    """FinetunedRePo.train_encoder, /root/reference/algorithms/repo/repo_adapt.py:26-94: the encoder alone is trained
    (its own Adam) on reward NLL + beta * (KL(post || prior) - target_kl) through the FROZEN filter and reward head; the
    dual variable follows as in RePo.  Parity status: PINNED by tests/golden/finetune_tiny.npz (the reference's class run
    by tests/golden/gen_golden.py)."""

    def __init__(self, cfg, action_size, params=None, seed=7):
        super().__init__(cfg, action_size, params=params, seed=seed)
        assert self.is_repo
        self.encoder_params = list(self.p["encoder"].values())
        self.encoder_opt = Adam(self.encoder_params, cfg.model_lr)

    def train_encoder(self, obs, actions, rewards, nonterms, eps_prior, eps_post, apply=True):
        c, p = self.c, self.p
        L, B = obs.shape[:2]
        T = L - 1
        embeds = encoder_fwd(p["encoder"], obs.reshape(L * B, *obs.shape[2:])).reshape(L, B, -1)
        frozen = list(p["transition_model"].values()) + list(p["reward_model"].values())
        for q in frozen:
            q.requires_grad_(False)
        try:
            beliefs, _, pm, ps, post_s, qm, qs = observe(p["transition_model"], torch.zeros(B, c.belief_size),
                                                         torch.zeros(B, c.state_size), actions[:-1], embeds[1:],
                                                         nonterms[:-1], eps_prior, eps_post)
            r_pred = scalar_head(p["reward_model"], beliefs.reshape(T * B, -1), post_s.reshape(T * B, -1)).reshape(T, B)
        finally:
            for q in frozen:
                q.requires_grad_(True)
        mask = nonterms[:-1].squeeze(-1)
        reward_loss = ((0.5 * (r_pred - rewards[:-1].squeeze(-1)) ** 2 + 0.5 * LOG_2PI) * mask).mean((0, 1))
        kl_div = normal_kl(qm, qs, pm, ps).sum(2).mean((0, 1))
        kl_viol = kl_div - c.target_kl
        kl_loss = self.log_beta.exp().detach() * kl_viol
        encoder_loss = reward_loss + kl_loss
        self.encoder_opt.zero_grad()
        encoder_loss.backward()
        self.last["encoder_grads"] = [q.grad.detach().clone() for q in self.encoder_params]
        self.last["encoder_total_norm"] = float(clip_grad_norm(self.encoder_params, c.grad_clip_norm))
        if apply:
            self.encoder_opt.step()
        beta_loss = -self.log_beta * kl_viol.detach()
        self.beta_opt.zero_grad()
        beta_loss.backward()
        if apply:
            self.beta_opt.step()
        out = {"train/reward_loss": reward_loss, "train/kl_loss": kl_loss, "train/kl_div": kl_div,
               "train/encoder_loss": encoder_loss, "train/beta": self.log_beta.exp(), "train/beta_loss": beta_loss}
        return {k: float(v.detach()) for k, v in out.items()}

    def update(self, obs_u8, actions, rewards, dones, noise):
        obs = torch.from_numpy(fx.preprocess_u8(np.asarray(obs_u8)))
        t = {k: torch.as_tensor(v) for k, v in noise.items()}
        return self.train_encoder(obs, torch.as_tensor(actions), torch.as_tensor(rewards), 1 - torch.as_tensor(dones),
                                  t["obs_prior"], t["obs_post"])


# --------------------------------------------------------------------------- multitask (task-conditioned) agents
# MultitaskDreamer / MultitaskRePo, /root/reference/algorithms/repo/dreamer_mt.py:28-301, repo_mt.py:13-135, with the
# conditioned modules of models/encoder.py:68-88, models/decoder.py:96-123,198-213, models/rssm.py:187-249,
# models/actor_critic.py:28-55,104-139.  share_repr=False (the default of experiments/train_repo.py:70).
# Parity status: PINNED on the reference's own classes (tests/golden/gen_golden.py: run_mt_case ->
# mt_dreamer_tiny.npz, mt_repo_tiny.npz; tests/test_oracle_golden.py).
def _film(p, cond, channels):
    g, b = F.linear(cond, p["film.weight"], p["film.bias"]).chunk(2, dim=1)
    return g.split(list(channels), dim=1), b.split(list(channels), dim=1)


def _mod(x, gamma, beta):
    return (1 + gamma[..., None, None]) * x + beta[..., None, None]


def cond_encoder_fwd(p, obs, cond):
    """ConditionalVisualEncoder.forward, encoder.py:78-88."""
    gs, bs = _film(p, cond, (32, 64, 128, 256))
    h = obs
    for i in range(4):
        h = F.relu(_mod(F.conv2d(h, p[f"conv{i + 1}.weight"], p[f"conv{i + 1}.bias"], stride=2), gs[i], bs[i]))
    return h.reshape(h.shape[0], -1)


def cond_decoder_fwd(p, belief, state, cond):
    """ConditionalVisualObservationModel.forward, decoder.py:111-123: fc1 on cat([belief, state]) -- the pixel decoder
    concatenates NOTHING, the condition enters through FiLM on conv1..conv3 only (conv4 is not modulated)."""
    gs, bs = _film(p, cond, (128, 64, 32))
    h = F.linear(torch.cat([belief, state], 1), p["fc1.weight"], p["fc1.bias"])
    h = h.view(-1, p["fc1.weight"].shape[0], 1, 1)
    for i in range(3):
        h = F.relu(_mod(F.conv_transpose2d(h, p[f"conv{i + 1}.weight"], p[f"conv{i + 1}.bias"], stride=2), gs[i], bs[i]))
    return F.conv_transpose2d(h, p["conv4.weight"], p["conv4.bias"], stride=2)


def cond_imagine(rssm, actor, belief0, state0, cond, horizon, eps_act, eps_prior):
    """ConditionalTransitionModel.imagine, rssm.py:221-249: the policy sees [belief | state | cond] (detached), the
    belief update the pseudo-action [action | cond]."""
    beliefs, states, means, stds = [], [], [], []
    belief, state = belief0, state0
    for t in range(horizon - 1):
        a_mean, a_std = actor_fwd(actor, belief.detach(), torch.cat([state.detach(), cond], 1))
        action = torch.tanh(a_mean + a_std * eps_act[t])
        belief = compute_belief(rssm, belief, state, torch.cat([action, cond], 1))
        state, pm, ps = gaussian_head(rssm, "fc_embed_belief_prior", "fc_state_prior", belief, eps_prior[t])
        beliefs.append(belief)
        states.append(state)
        means.append(pm)
        stds.append(ps)
    return [torch.stack(x, 0) for x in (beliefs, states, means, stds)]


class OracleMultitask(OracleAgent):
    """cfg.algo: "dreamer_multitask" or "repo_multitask"; num_tasks = C.  Batches carry tasks (L, B, C) first."""

    def __init__(self, cfg, action_size, num_tasks, params=None, seed=7):
        np_params = params if params is not None else fx.make_params(action_size, seed, cond=num_tasks)
        super().__init__(cfg, action_size, params=np_params)
        self.C = num_tasks
        self.is_repo = cfg.algo == "repo_multitask"
        if self.is_repo:   # one dual variable per task (repo_mt.py:24-32)
            self.log_beta = torch.full((num_tasks,), float(np.log(cfg.init_beta)), dtype=torch.float32, requires_grad=True)
            self.beta_opt = Adam([self.log_beta], cfg.beta_lr)

    def train_dynamics(self, tasks, obs, actions, rewards, nonterms, eps_prior, eps_post, apply=True):
        c, p = self.c, self.p
        L, B = obs.shape[:2]
        T = L - 1
        embeds = cond_encoder_fwd(p["encoder"], obs.reshape(L * B, *obs.shape[2:]), tasks.reshape(L * B, -1)).reshape(L, B, -1)
        b0 = torch.zeros(B, c.belief_size)
        s0 = torch.zeros(B, c.state_size)
        pseudo = torch.cat((actions[:-1], tasks[:-1]), dim=2)
        beliefs, prior_s, pm, ps, post_s, qm, qs = observe(
            p["transition_model"], b0, s0, pseudo, embeds[1:], nonterms[:-1], eps_prior, eps_post)
        tk = tasks[1:].reshape(T * B, -1)   # "Match task timestep" (dreamer_mt.py:186-187)
        fb, fs = beliefs.reshape(T * B, -1), post_s.reshape(T * B, -1)
        if self.is_repo:
            recon = cond_decoder_fwd(p["obs_model"], fb.detach(), fs.detach(), tk)
        else:
            recon = cond_decoder_fwd(p["obs_model"], fb, fs, tk)
        recon = recon.reshape(T, B, *obs.shape[2:])
        obs_loss = (0.5 * (recon - obs[1:]) ** 2 + 0.5 * LOG_2PI).sum((2, 3, 4)).mean((0, 1))
        r_pred = scalar_head(p["reward_model"], fb, torch.cat([fs, tk], 1)).reshape(T, B)
        mask = nonterms[:-1].squeeze(-1)
        reward_loss = ((0.5 * (r_pred - rewards[:-1].squeeze(-1)) ** 2 + 0.5 * LOG_2PI) * mask).mean((0, 1))
        out = {}
        if self.is_repo:
            kl_prior = normal_kl(qm.detach(), qs.detach(), pm, ps).sum(2)
            kl_post = normal_kl(qm, qs, pm.detach(), ps.detach()).sum(2)
            alpha = c.prior_train_steps / (1 + c.prior_train_steps)
            kl_div = alpha * kl_prior + (1 - alpha) * kl_post          # (T, B): per row, repo_mt.py:86-88
            kl_viol = kl_div - c.target_kl
            log_beta = tasks[1:] @ self.log_beta                       # (T, B)
            kl_loss = (log_beta.exp().detach() * kl_viol).mean()
            out["train/kl_div"] = kl_div.mean()
        else:
            kl = normal_kl(qm, qs, pm, ps).sum(2)
            kl_loss = torch.max(kl, torch.full((1,), float(c.free_nats))).mean((0, 1))
        model_loss = obs_loss + reward_loss + kl_loss
        self.model_opt.zero_grad()
        model_loss.backward()
        self.last["model_grads"] = [None if q.grad is None else q.grad.detach().clone() for q in self.model_params]
        self.last["model_total_norm"] = float(clip_grad_norm(self.model_params, c.grad_clip_norm))
        if apply:
            self.model_opt.step()
        out.update({"train/obs_loss": obs_loss, "train/reward_loss": reward_loss, "train/kl_loss": kl_loss,
                    "train/model_loss": model_loss})
        if self.is_repo:
            beta_loss = -(log_beta * kl_viol.detach()).mean()
            self.beta_opt.zero_grad()
            beta_loss.backward()
            if apply:
                self.beta_opt.step()
            out["train/beta_loss"] = beta_loss
            for i in range(self.C):
                out[f"train/beta_{i}"] = self.log_beta[i].exp()
        self.last["observe"] = [x.detach() for x in (beliefs, prior_s, pm, ps, post_s, qm, qs)]
        return beliefs.detach(), post_s.detach(), {k: float(v.detach()) for k, v in out.items()}

    def train_actor_critic(self, tasks, beliefs, post_states, eps_act, eps_prior, eps_ent, apply=True):
        """tasks (N, C).  dreamer_mt.py:230-301."""
        c, p = self.c, self.p
        H = c.horizon
        frozen = self.model_params + self.value_params
        for q in frozen:
            q.requires_grad_(False)
        try:
            ib, istate, im, isd = cond_imagine(p["transition_model"], p["actor_model"], beliefs, post_states, tasks, H,
                                               eps_act, eps_prior)
            Hm, N = ib.shape[:2]
            tk = tasks[None].repeat(Hm, 1, 1).reshape(Hm * N, -1)
            fb, fs = ib.reshape(Hm * N, -1), istate.reshape(Hm * N, -1)
            fsc = torch.cat([fs, tk], 1)
            r_pred = scalar_head(p["reward_model"], fb, fsc).reshape(Hm, N)
            v_pred = scalar_head(p["value_model"], fb, fsc).reshape(Hm, N)
        finally:
            for q in frozen:
                q.requires_grad_(True)
        a_mean, a_std = actor_fwd(p["actor_model"], fb, fsc)
        action_entropy = tanh_normal_entropy(a_mean, a_std, eps_ent).mean()
        latent_entropy = isd.log().sum(-1).mean()   # dreamer_mt.py:258: WITHOUT the Normal entropy's constant
        disc = c.gamma * torch.ones_like(r_pred)
        returns = lambda_return(r_pred[:-1], v_pred[:-1], disc[:-1], v_pred[-1], c.gae_lambda)
        actor_loss = -returns.mean() - c.action_ent_coef * action_entropy - c.latent_ent_coef * latent_entropy
        self.actor_opt.zero_grad()
        for q in self.model_params + self.value_params:
            q.grad = None
        actor_loss.backward()
        self.last["actor_grads"] = [q.grad.detach().clone() for q in self.actor_params]
        self.last["actor_total_norm"] = float(clip_grad_norm(self.actor_params, c.grad_clip_norm))
        if apply:
            self.actor_opt.step()
        nv = (Hm - 1) * N
        v = scalar_head(p["value_model"], fb[:nv].detach(), fsc[:nv].detach())
        value_loss = (0.5 * (v - returns.detach().reshape(-1)) ** 2 + 0.5 * LOG_2PI).mean()
        self.value_opt.zero_grad()
        value_loss.backward()
        self.last["value_grads"] = [q.grad.detach().clone() for q in self.value_params]
        self.last["value_total_norm"] = float(clip_grad_norm(self.value_params, c.grad_clip_norm))
        if apply:
            self.value_opt.step()
        self.last["imagine"] = [x.detach() for x in (ib, istate, im, isd)]
        return {"train/actor_loss": float(actor_loss.detach()), "train/value_loss": float(value_loss.detach()),
                "train/action_entropy": float(action_entropy.detach()),
                "train/latent_entropy": float(latent_entropy.detach())}

    def update(self, tasks, obs_u8, actions, rewards, dones, noise):
        obs = torch.from_numpy(fx.preprocess_u8(np.asarray(obs_u8)))
        tk = torch.as_tensor(tasks)
        t = {k: torch.as_tensor(v) for k, v in noise.items()}
        beliefs, post, scal = self.train_dynamics(tk, obs, torch.as_tensor(actions), torch.as_tensor(rewards),
                                                  1 - torch.as_tensor(dones), t["obs_prior"], t["obs_post"])
        scal.update(self.train_actor_critic(tk[1:].flatten(0, 1), beliefs.flatten(0, 1), post.flatten(0, 1),
                                            t["img_act"], t["img_prior"], t["entropy"]))
        return beliefs, post, scal
