"""CPU restatement of Transgo's live search (WP_MCTS + Node_V).  TEST INFRASTRUCTURE ONLY.

Parity oracle for the HIP tree kernels.  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may
import this module; the product path (transgo_amd/) never does.

Follows /root/reference/self_play.py:51-95 (Node_V) and :575-875 (WP_MCTS) -- cited per method -- but is organised as
the three phases the GPU engine runs per wave (collect -> evaluate -> absorb) and draws randomness from an explicit
per-game `np.random.RandomState` (the reference uses the global legacy stream; `np.random.seed(s)` + global calls are
the same MT19937 stream as `RandomState(s)` methods).

Bit-exactness depends on NumPy-2 (NEP 50) scalar promotion; every arithmetic statement below keeps the operand kinds
of the reference statement it restates (SURVEY.md §8a Note N):
  * policy/value arrive as np.float32; priors set in `absorb` are f32/f32 -> f32; after root noise they are f64.
  * value_sum is f32 accumulated leaf->root; terminal results are Python ints.
  * var / mean are Python floats until the first backup through a node, f32 afterwards.
  * the PUCT score is evaluated in f64; the first maximal child in ascending-action order wins `max`, and the tie list
    is rebuilt with `==` before `RandomState.choice`.
Pinned against golden vectors recorded from the imported reference: tests/golden/search_*.npz.
"""
import math

import numpy as np


class Vertex:
    """Node_V (self_play.py:51-95)."""
    __slots__ = ("prior", "state", "n", "pending", "w", "mean", "var", "kids", "open")

    def __init__(self, prior):
        self.prior = prior
        self.state = None
        self.n = 0            # total_visit_count
        self.pending = 0      # ons: WU-UCT unobserved samples
        self.w = 0            # value_sum
        self.mean = 0.        # value_mean (Python float until first backup)
        self.var = 0.         # value_var
        self.kids = {}        # action -> Vertex, insertion order = ascending legal action
        self.open = False     # real_expanded

    def q(self):                                         # self_play.py:67-68
        return self.w / (self.n + 1)

    def spawn(self, action_priors, value=0.0):           # self_play.py:70-77
        for a, p in action_priors.items():
            k = Vertex(p)
            k.w = -1 * value
            self.kids[a] = k

    def absorb_value(self, value):                       # self_play.py:84-88
        t = self.mean
        self.mean = self.q()
        self.var = self.var + (value - t) * (value - self.mean)

    def add_root_noise(self, rng):                       # self_play.py:90-95
        acts = list(self.kids.keys())
        noise = rng.dirichlet([0.03] * len(acts))
        for a, e in zip(acts, noise):
            self.kids[a].prior = self.kids[a].prior * (1 - 0.25) + e * 0.25


def temperature(game_step):                              # configure.py:75-79
    return 0.65 + (1.0 - 0.65) * math.exp(-1. * game_step / 10)


class OracleSearch:
    """WP_MCTS (self_play.py:575-875).  `evaluate(obs f32[k,C,S,S]) -> (policy f32[k,A], value f32[k,1])`."""

    def __init__(self, env, evaluate, rng, num_simulation=210, parallel_readouts=4, c1=3, c2=0.05, wu_loss=2,
                 board_size=9, trace=None):
        self.env, self.evaluate, self.rng = env, evaluate, rng
        self.num_simulation, self.readouts = num_simulation, parallel_readouts
        self.c1, self.c2, self.wu = c1, c2, wu_loss
        self.S = board_size
        self.A = board_size * board_size + 1
        self.trace = trace                     # optional list collecting (kind, payload) events for tests
        self.sims_done = 0
        self.leaves_evaluated = 0
        self.reset_root()

    # -- evaluation: self_play.py:796-829 with the dead sub_model branch removed (init_sub_model=False) --------------
    def _eval(self, states):
        obs = np.array([self.env.encode(s) for s in states], dtype="float32")
        policy, value = self.evaluate(obs)
        self.leaves_evaluated += len(states)
        return policy, value

    def _open_root(self, root):                          # self_play.py:599-605 / :864-870 (raw, un-normalised priors)
        policy, value = self._eval([root.state])
        policy, value = policy[0], value[0][0]
        legal = self.env.getLegalAction(root.state)
        root.spawn({i: p for i, p in enumerate(policy) if i in legal}, value)
        root.open = True

    def reset_root(self):                                # self_play.py:595-605
        self.root = Vertex(0)
        self.root.state, _ = self.env.reset()
        self._open_root(self.root)

    # -- selection: self_play.py:706-725 ------------------------------------------------------------------------------
    def score(self, parent, child):
        u = self.c1 * child.prior * np.sqrt(parent.n + parent.pending) / (child.n + child.pending + 1)
        v = np.clip(child.var, 0, 3)
        s = self.c2 * np.sqrt(1 + v)
        return u + s + (-child.q())

    def pick_child(self, node):
        best = max(self.score(node, c) for c in node.kids.values())
        tied = [a for a, c in node.kids.items() if self.score(node, c) == best]
        a = self.rng.choice(tied)
        return a, node.kids[a]

    # -- one wave: self_play.py:607-654 -------------------------------------------------------------------------------
    def collect(self):
        """Selection + leaf stepping for up to `readouts` paths (<= 2*readouts attempts).  Terminal leaves are backed
        up immediately and do not count as paths."""
        paths, leaves = [], []
        attempts = 0
        while len(paths) < self.readouts and attempts < self.readouts * 2:
            node = self.root
            path = [node]
            attempts += 1
            while node.open:
                act, node = self.pick_child(node)
                path.append(node)
            leaf_state, done = self.env.step(path[-2].state, act)
            node.state = leaf_state
            if done:                                     # self_play.py:638-642
                value = 1 if self.env.getPlayer(node.state) == self.env.getWinner(node.state) else -1
                self.backup(path, value)
                continue
            legal = self.env.getLegalAction(leaf_state)
            node.kids = {}                               # expand() overwrites every legal key; same key set each time
            node.spawn({i: 0.0 for i in legal})
            for v in reversed(path):                     # incomplete_update, self_play.py:767-770
                v.pending += self.wu
            paths.append(path)
            leaves.append(leaf_state)
        return paths, leaves

    def absorb(self, paths, leaves, probs, values):      # self_play.py:651-654, :727-755, :772-774
        for path, leaf_state, prob, value in zip(paths, leaves, probs, values):
            for v in reversed(path):
                v.pending -= self.wu
            leaf = path[-1]
            if leaf.open:                                # duplicate leaf inside one wave: simulation dropped
                continue
            value = value[0]
            legal = self.env.getLegalAction(leaf_state)
            scale = sum(prob[legal])
            if scale > 0:
                for a in legal:
                    leaf.kids[a].prior = prob[a] / scale
                    leaf.kids[a].w = -1 * value
            leaf.open = True
            self.backup(path, value)

    def backup(self, path, value):                       # self_play.py:758-764
        for v in reversed(path):
            v.w += value
            v.n += 1
            v.absorb_value(value)
            value = -value
        self.sims_done += 1

    def wave(self):
        paths, leaves = self.collect()
        if paths:
            probs, values = self._eval(leaves)
            self.absorb(paths, leaves, probs, values)

    # -- one move: self_play.py:657-687 -------------------------------------------------------------------------------
    def search_move(self, selfplay=True):
        if selfplay:
            self.root.add_root_noise(self.rng)
        n0 = self.root.n
        while self.root.n < n0 + self.num_simulation:
            self.wave()
        counts = np.array([self.root.kids[a].n if a in self.root.kids else 0 for a in range(self.A)])
        counts = np.where(counts == 1, 0, counts)
        pi = counts / np.sum(counts)
        tau = temperature(self.env.getStep(self.root.state)) if selfplay else 0.12
        powed = np.power(counts, 1.0 / tau)
        probs = np.array(powed) / np.sum(powed)
        action = self.rng.choice(np.arange(self.A), p=probs)
        obs = self.env.encode(self.root.state)
        return action, pi, obs, dict(n0=n0, counts=counts, tau=tau)

    def select_action(self, state):                      # self_play.py:689-703
        self.root = Vertex(0)
        self.root.state = state
        self._open_root(self.root)
        action, _, _, _ = self.search_move(selfplay=False)
        return action

    def advance(self, action):                           # update_with_action, self_play.py:857-872
        nxt, done = self.env.step(self.root.state, action)
        self.root = self.root.kids[action]
        if not self.root.open:
            self.root.state = nxt
            self._open_root(self.root)
        return done


def targets_for_game(env, final_state, observations, pis, players, board_size=9):
    """Post-game target generation + 8-fold augmentation in the reference's append order (self_play.py:929-967).
    Returns the list of (obs f32[C,S,S], pi f64[A], z float, own f64[S*S]) tuples."""
    S = board_size
    z = np.zeros(len(players))
    winner = env.getWinner(final_state)
    z[np.array(players) == winner] = 1
    z[np.array(players) != winner] = -1
    _, terr = env.getScoreAndTerritory(final_state)
    own = np.zeros((len(players), S * S))
    own[np.array(players) == 1] = terr
    own[np.array(players) != 1] = -1 * terr
    out = []
    for ob, pi, zz, ow in zip(observations, pis, z, own):
        for i in (1, 2, 3, 4):
            board_p, pass_p = pi[:-1], pi[-1]
            rp = np.rot90(board_p.reshape(S, S), i)
            ro = np.array([np.rot90(pl, i) for pl in ob])
            rw = np.rot90(ow.reshape(S, S), i)
            out.append((ro, np.append(rp.flatten(), pass_p), zz, rw.flatten()))
            fo = np.array([np.fliplr(pl) for pl in ro])
            out.append((fo, np.append(np.fliplr(rp).flatten(), pass_p), zz, np.fliplr(rw).flatten()))
    return out


def policy_evaluate(env, evaluate_train, evaluate_eval, n_games, seed, num_simulation=210, board_size=9, shared_stream=False,
                    return_rng=False):
    """SelfPlay.policy_evaluate (self_play.py:986-1040): n_games between the train agent and the evaluation agent, the train
    agent's colour alternating from BLACK, every move by select_action; both agents draw from one stream (the reference's
    global np.random).  The reference never seeds; the parity harness seeds the stream with seed + i at the start of game i
    (what the batched engine does: its games run concurrently, so each owns a stream).  shared_stream=True is the reference's own
    form -- ONE stream seeded once and carried from game to game -- which tests/golden/policy_evaluate.json (recorded from the
    reference's policy_evaluate itself) pins.  Returns (winners[n_games], colours[n_games], win_ratio[, the stream])."""
    BLACK, WHITE = 1, 2
    color = BLACK
    winners, colours = [], []
    one = np.random.RandomState(int(seed) % (2 ** 32)) if shared_stream else None
    rng = one
    for i in range(n_games):
        if not shared_stream:
            rng = np.random.RandomState(int(seed + i) % (2 ** 32))
        train = OracleSearch(env, evaluate_train, rng, num_simulation=num_simulation, board_size=board_size)
        evalu = OracleSearch(env, evaluate_eval, rng, num_simulation=num_simulation, board_size=board_size)
        bots = {BLACK: train, WHITE: evalu} if color == BLACK else {BLACK: evalu, WHITE: train}
        state, done = env.reset()
        while not done:
            action = bots[env.getPlayer(state)].select_action(state)
            state, done = env.step(state, action)
        winners.append(env.getWinner(state)); colours.append(color)
        color = BLACK + WHITE - color
    winners, colours = np.array(winners), np.array(colours)
    ratio = float((winners == colours).sum()) / n_games
    return (winners, colours, ratio, rng) if return_rng else (winners, colours, ratio)
