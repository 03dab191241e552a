"""CPU stand-in for ptmcmc_amd.parallel.EngineShard, built on the oracle -- TEST INFRASTRUCTURE.

Implements the five backend methods of ShardedLadder with numpy/torch-CPU buffers so that the multi-rank protocol
(halo messages, windowed decision replay, boundary rows) can be exercised over gloo without a GPU.  The exchange
decisions restate parallel_tempering_chains::step's swap phase (chain.cc:1410-1537) on the shard's llike window."""
import ctypes as C

import numpy as np
import torch

import oracle_lib as O


class OracleShard:
    def __init__(self, lad, r0, nloc, seed):
        """lad: an oracle Ladder over the GLOBAL ladder (only rungs [r0, r0+nloc) of it are kept valid)"""
        self.lad, self.r0, self.nloc, self.seed = lad, r0, nloc, seed
        self.W, self.Nt, self.D = lad.W, lad.Nt, lad.D
        self.row_doubles = self.W * (self.D + 2)
        s = lad.s.contents
        N = self.Nt * self.W
        self.x = np.ctypeslib.as_array(s.x, shape=(N, self.D))
        self.ll = np.ctypeslib.as_array(s.llike, shape=(N,))
        self.lp = np.ctypeslib.as_array(s.lprior, shape=(N,))
        self.nhist = np.ctypeslib.as_array(s.nhist, shape=(N,))
        self.nsize = np.ctypeslib.as_array(s.nsize, shape=(N,))
        self.touched = np.ctypeslib.as_array(s.touched, shape=(N,))
        self.swap_try = np.zeros((self.W, self.Nt - 1), dtype=np.int64)
        self.swap_acc = np.zeros((self.W, self.Nt - 1), dtype=np.int64)
        self.ms = s.maxswaps
        self.thresh = (self.Nt - 1) * s.swap_rate / self.ms
        self.moves = []
        self.far = False

    def alloc(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def idx(self, w, r):
        return w * self.Nt + r            # oracle chain order

    def copy_llike(self, first, n, dst):
        out = dst.numpy().reshape(n, self.W)
        for k in range(n):
            for w in range(self.W):
                out[k, w] = self.ll[self.idx(w, self.r0 + first + k)]

    def _add_state(self, c):
        every = self.lad.s.contents.add_every_N
        if self.nhist[c] % every == 0:
            self.nsize[c] += 1
        self.nhist[c] += 1

    # ---- recovery of runs longer than the halo (ptm_set_shard_map / ptm_exchange_redo_count / ptm_exchange_redo)
    def set_shard_map(self, sizes, halo):
        ends = np.cumsum(sizes)
        self.blind_spans = []             # per inner boundary: the picks that, all surviving, blind the shard below it
        for k in range(len(sizes) - 1):
            B, h = int(ends[k]), min(halo, sizes[k + 1])
            self.blind_spans.append(range(B - 1, B + h))
        self.redo_ws = []

    def exchange_redo_count(self):
        return len(self.redo_ws)

    def exchange_redo(self, ll_all, lp_all, send_up, send_down):
        """the ladders exchange_decide left alone, decided on the whole ladder's llikes"""
        Nt, W, D = self.Nt, self.W, self.D
        la = ll_all.numpy().reshape(Nt, W)
        su = None if send_up is None else send_up.numpy().reshape(D + 2, W)
        sd = None if send_down is None else send_down.numpy().reshape(D + 2, W)
        for w, cand, lu in self.redo_ws:
            own = lambda r, w=w: self.ll[self.idx(w, r)] if self.r0 <= r < self.r0 + self.nloc else la[r, w]
            self._decide_ladder(w, cand, lu, 0, Nt - 1, own, su, sd)
        self.redo_ws = []
        self._apply_moves()

    def _draws(self, w, step):
        Nt = self.Nt
        cand, lu = [], []
        for k in range(self.ms):
            o = O.draw_block(self.seed, 1, w, step, k)
            n = -2
            if Nt > 1 and O.lib().ptmo_u01(o[0]) < self.thresh:
                n = int(O.lib().ptmo_u01(o[1]) * (Nt - 1))
                for j in cand:
                    if j == n or j + 1 == n:
                        n = -2
                        break
            cand.append(n)
            lu.append(O.lib().ptmo_log(O.lib().ptmo_u01(o[2])))
        return cand, lu

    def _apply_moves(self):
        # rows that stay inside the shard move now (the engine's move kernel); arrivals wait for install()
        for c, x, ll, lp in self.moves:
            self.x[c] = x; self.ll[c] = ll; self.lp[c] = lp
        self.moves = []

    def exchange_decide(self, ll_below, ll_above, H, send_up, send_down):
        r0, r1, Nt, W, D = self.r0, self.r0 + self.nloc, self.Nt, self.W, self.D
        step = self.lad.step
        wlo = r0 - (1 if ll_below is not None else 0)
        whi = r1 - 1 + (H if ll_above is not None else 0)
        lb = None if ll_below is None else ll_below.numpy()
        la = None if ll_above is None else ll_above.numpy().reshape(H, W)
        su = None if send_up is None else send_up.numpy().reshape(D + 2, W)
        sd = None if send_down is None else send_down.numpy().reshape(D + 2, W)
        self.touched[:] = 0
        self.moves = []        # (dst chain, row) applied in finish
        self.arrive = []       # (dst chain, "above"/"below", w)
        self.redo_ws = []
        for w in range(W):
            cand, lu = self._draws(w, step)
            alive = set(i for i in cand if i >= 0)
            if any(all(i in alive for i in span) for span in getattr(self, "blind_spans", [])):
                self.redo_ws.append((w, cand, lu))      # some shard cannot decide this ladder from its halo: all leave it alone
                continue

            def llike0(r, w=w):
                if r < r0:
                    return lb[w]
                if r >= r1:
                    return la[r - r1, w]
                return self.ll[self.idx(w, r)]
            self._decide_ladder(w, cand, lu, wlo, whi, llike0, su, sd)
        self._apply_moves()

    def _decide_ladder(self, w, cand, lu, wlo, whi, llike0, su, sd):
        r0, r1, Nt, D = self.r0, self.r0 + self.nloc, self.Nt, self.D
        beta = self.lad.beta
        if True:
            cur = {}      # rung -> (llike, source rung) view of the window
            def get(r):
                if r not in cur:
                    cur[r] = (llike0(r), r)
                return cur[r]
            taint = Nt + 1
            tch = {}
            for k, i in enumerate(cand):
                if i < 0:
                    continue
                if i + 1 > whi or i < wlo:
                    if i == whi and i + 1 < Nt:
                        taint = i
                    continue
                if i + 1 >= taint:
                    if i + 1 <= r1:
                        self.far = True
                    taint = min(taint, i)
                    continue
                (lla, sa), (llb, sb) = get(i), get(i + 1)
                a_, b_ = (lla if lla > -1e200 else -1e200), (llb if llb > -1e200 else -1e200)
                logH = -(beta[i + 1] - beta[i]) * (b_ - a_)
                acc = True
                if logH < 0:
                    acc = lu[k] < logH
                if acc:
                    if i + 1 == r0 and sd is not None:         # the row leaving downwards
                        s = sb
                        if not (r0 <= s < r1):
                            self.far = True
                        else:
                            c = self.idx(w, s)
                            sd[:D, w] = self.x[c]; sd[D, w] = self.ll[c]; sd[D + 1, w] = self.lp[c]
                    if i + 1 == r1 and r1 < Nt and su is not None:
                        c = self.idx(w, i)
                        su[:D, w] = self.x[c]; su[D, w] = self.ll[c]; su[D + 1, w] = self.lp[c]
                    cur[i], cur[i + 1] = (llb, sb), (lla, sa)
                    if r0 <= i < r1:
                        self.swap_acc[w, i] += 1
                tch[i] = tch.get(i, 0) + 1
                tch[i + 1] = tch.get(i + 1, 0) + 1
                if r0 <= i < r1:
                    self.swap_try[w, i] += 1
            for r, n in tch.items():
                if r0 <= r < r1:
                    c = self.idx(w, r)
                    self.touched[c] = n
                    _, s = cur[r]
                    if r0 <= s < r1:
                        if s != r:
                            cs = self.idx(w, s)
                            self.moves.append((c, self.x[cs].copy(), self.ll[cs], self.lp[cs]))
                    else:
                        self.arrive.append((c, "above" if s >= r1 else "below", w))

    can_overlap = True

    # ---- evolving ladders: the gathered form (ShardedLadder.step_gathered).  The stand-in runs the oracle's own exchange phase
    #      on its replica of the WHOLE ladder: the other shards' llikes / lpriors come from the gathered views, their rows are
    #      tags (NaN, rung number in the first entry) -- a tag that lands on an own rung names the neighbour a row arrives from,
    #      a real row that lands on a foreign rung is one that leaves.
    @property
    def gathered(self):
        return self.lad.s.contents.evolve_rate > 0

    needs_lprior = True

    def copy_lprior(self, first, n, dst):
        out = dst.numpy().reshape(n, self.W)
        for k in range(n):
            for w in range(self.W):
                out[k, w] = self.lp[self.idx(w, self.r0 + first + k)]

    @staticmethod
    def sub(buf, off, n):
        return buf[off:off + n]

    @staticmethod
    def dcopy(dst, src):
        dst.copy_(src)

    def exchange_decide_gathered(self, ll_all, lp_all, send_up, send_down):
        r0, r1, Nt, W, D = self.r0, self.r0 + self.nloc, self.Nt, self.W, self.D
        assert D >= 2
        la = ll_all.numpy().reshape(Nt, W)
        lpa = None if lp_all is None else lp_all.numpy().reshape(Nt, W)
        su = None if send_up is None else send_up.numpy().reshape(D + 2, W)
        sd = None if send_down is None else send_down.numpy().reshape(D + 2, W)
        for w in range(W):
            for r in list(range(0, r0)) + list(range(r1, Nt)):
                c = self.idx(w, r)
                self.ll[c] = la[r, w]
                self.lp[c] = 0.0 if lpa is None else lpa[r, w]
                self.x[c, :] = np.nan
                self.x[c, 0] = r
        s = self.lad.s.contents
        st = np.ctypeslib.as_array(s.swap_count, shape=(W, Nt - 1))
        sa = np.ctypeslib.as_array(s.swap_accept_count, shape=(W, Nt - 1))
        t0, a0 = st.copy(), sa.copy()
        O.lib().ptmo_exchange_phase(self.lad.s, self.lad.rng)        # touch counts, add_state counts, temperatures: in place
        self.swap_try[:, r0:min(r1, Nt - 1)] += (st - t0)[:, r0:min(r1, Nt - 1)]
        self.swap_acc[:, r0:min(r1, Nt - 1)] += (sa - a0)[:, r0:min(r1, Nt - 1)]
        self.counted = True                                           # (the phase counted the touched rungs' add_state calls itself)
        self.arrive = []
        for w in range(W):
            for r in range(r0, r1):
                c = self.idx(w, r)
                if np.isnan(self.x[c, -1]):
                    src = int(self.x[c, 0])
                    assert src == r0 - 1 or r1 <= src, src
                    self.arrive.append((c, "above" if src >= r1 else "below", w))
            for r in list(range(0, r0)) + list(range(r1, Nt)):
                c = self.idx(w, r)
                if not np.isnan(self.x[c, -1]):                       # an own row that left
                    buf = su if r >= r1 else sd
                    buf[:D, w] = self.x[c]; buf[D, w] = self.ll[c]; buf[D + 1, w] = self.lp[c]

    def install(self, recv_below, recv_above):
        """rows that arrived from the neighbours land in the holes the decisions named"""
        D, W = self.D, self.W
        rb = None if recv_below is None else recv_below.numpy().reshape(D + 2, W)
        ra = None if recv_above is None else recv_above.numpy().reshape(D + 2, W)
        for c, where, w in self.arrive:
            buf = ra if where == "above" else rb
            self.x[c] = buf[:D, w]; self.ll[c] = buf[D, w]; self.lp[c] = buf[D + 1, w]
        self.arrive = []

    def sweep_rungs(self, first, n, closes_step):
        L = O.lib()
        lad = self.lad
        for w in range(self.W):
            for r in range(self.r0 + first, self.r0 + first + n):
                c = self.idx(w, r)
                if self.touched[c]:
                    if not getattr(self, "counted", False):
                        for _ in range(int(self.touched[c])):
                            self._add_state(c)
                    self.touched[c] = 0
                    continue
                L.ptmo_mh_step(lad.s, lad.pb.p, C.byref(lad._props[r]), lad.rng, w, r)
        if closes_step:
            lad.s.contents.step += 1

    def finish_and_sweep(self, recv_below, recv_above):
        self.install(recv_below, recv_above)
        self.sweep_rungs(0, self.nloc, True)

    def sync(self):
        if self.far:
            raise RuntimeError("halo exceeded / far move")

    # local results in engine order (rung-major over the local rungs)
    def local(self, arr):
        a = np.asarray(arr)
        return np.stack([a[self.idx(w, r)] for r in range(self.r0, self.r0 + self.nloc) for w in range(self.W)])
