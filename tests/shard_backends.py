"""Test-only shard backend: the same stage interface as badger_amcl_amd.sharded.HipShardBackend,
implemented with the CPU oracle, so that ShardedFilter's exchange logic (rank-ordered totals,
CDF slices, window assembly, even re-split) runs under gloo with world_size 2 and no GPU."""
import numpy as np
import torch

MASK48 = (1 << 48) - 1
LCG_A, LCG_C = 0x5DEECE66D, 0xB


def lcg_skip(state, n):
    a, c = 1, 0
    pa, pc = LCG_A, LCG_C
    while n:
        if n & 1:
            a, c = (a * pa) & MASK48, (c * pa + pc) & MASK48
        pc = (pc * pa + pc) & MASK48
        pa = (pa * pa) & MASK48
        n >>= 1
    return (a * state + c) & MASK48


class OracleShardBackend:
    device = "cpu"

    def __init__(self, orc, omap, planar, samples, min_samples, max_samples_global, seed, alpha=(0.0, 0.0)):
        self.orc, self.omap, self.planar = orc, omap, planar
        self.samples = np.ascontiguousarray(samples, dtype=np.float64).copy()
        self.pfh = orc.ParticleFilter(min_samples, max_samples_global, alpha[0], alpha[1], 85.0, seed=seed)
        self._max = max_samples_global
        self._total = torch.zeros(1, dtype=torch.float64)
        self._sum = torch.zeros(1, dtype=torch.float64)
        self.tree = None
        self.leaf = self.bins = 0
        self.conv = 0

    def n_local(self):
        return self.samples.shape[0]

    def max_samples(self):
        return self._max

    def score(self, data):
        ranges, angles, range_max = data
        if self.samples.shape[0]:
            self.orc.planar_apply(self.planar, self.omap, self.samples, ranges, angles, range_max, 0)
        t = 0.0
        for w in self.samples[:, 3]:
            t += w
        self._total[0] = t

    def local_total(self):
        return self._total

    def normalize(self, totals, global_n):
        T = 0.0
        for v in totals.tolist():
            T += v
        pf = self.pfh.pf
        if T > 0.0:
            self.samples[:, 3] /= T
            w_avg = T / global_n
            pf.w_slow = w_avg if pf.w_slow == 0.0 else pf.w_slow + pf.alpha_slow * (w_avg - pf.w_slow)
            pf.w_fast = w_avg if pf.w_fast == 0.0 else pf.w_fast + pf.alpha_fast * (w_avg - pf.w_fast)
        else:
            self.samples[:, 3] = 1.0 / global_n

    def build_cdf(self, flags):
        flags[0] = 0
        c = np.zeros(self.samples.shape[0] + 1)
        run = 0.0
        for i, w in enumerate(self.samples[:, 3]):
            run = run + w
            c[i + 1] = run
        self.cdf = c
        self._sum[0] = c[-1]

    def local_sum(self):
        return self._sum

    _resample_model = 0

    def resample_model(self):
        return self._resample_model

    def begin_resample(self, rng, leaf_count):
        pf = self.pfh.pf
        w_diff = 1.0 - pf.w_fast / pf.w_slow if pf.w_slow != 0.0 else float("nan")
        if not w_diff >= 0.0:
            w_diff = 0.0
        self._w_diff, self._rng0, self._chain, self._n_random = w_diff, rng, None, 0
        count = self.pfh.resample_limit(leaf_count)
        if w_diff > 0.0:
            if self._resample_model == 1:
                count = min(int(count * (1.0 + w_diff)), self._max)
                self._n_random = int(w_diff * count)
            else:
                # where every candidate draw finds its stream elements (serial walk; the product resolves it in
                # parallel): (position of the test element, random?)
                chain, q = [], 1
                for _ in range(self._max + 1):
                    rnd = lcg_skip(rng, q) / float(1 << 48) < w_diff
                    chain.append((q, rnd))
                    q += 3 if rnd else 2
                self._chain = chain
        return w_diff, count

    def end_resample(self, m):
        if self._resample_model == 1:
            consumed = 1 + 2 * self._n_random
        elif self._chain is not None:
            consumed = self._chain[m][0] - 1
        else:
            consumed = 2 * m
        if self._w_diff > 0.0:
            self.pfh.pf.w_slow = self.pfh.pf.w_fast = 0.0
        return lcg_skip(self._rng0, consumed)

    def _random_pose(self, state_before_first):
        """Node::randomFreeSpacePose from the two stream elements after `state_before_first`."""
        import ctypes as C
        st = C.c_uint64(state_before_first)
        pose = np.zeros(3)
        self.orc.lib().orc_random_free_space_pose(C.byref(self.pfh._free), C.byref(st),
                                                  pose.ctypes.data_as(C.POINTER(C.c_double)))
        return pose

    def resample_limit(self, leaf_count):
        return self.pfh.resample_limit(leaf_count)

    def systematic_window(self, rng, count, sums, sums_are_totals, rank, world, window, flags):
        start = lcg_skip(rng, 1) / float(1 << 48)
        n_random = self._n_random
        delta = 1.0 / (count - n_random)
        targets, t = [None] * n_random, start
        for _ in range(count - n_random):
            targets.append(t)
            t += delta
            if t > 1.0:
                t -= 1.0
        random = {m: lcg_skip(rng, 2 * m + 1) for m in range(n_random)}  # state before elements 2m+2, 2m+3
        self._fill_window(targets, 0, sums, sums_are_totals, rank, world, window, flags, random)

    def kld_insert(self, keys, n):
        k = keys.numpy()
        for q in range(n):
            self.tree.insert_key(k[:, q].astype(np.int32), 1.0)

    def kld_insert_window(self, window, n):
        self.kld_insert(window[3:6].contiguous(), n)

    def local_pose_keys(self):
        s = self.samples
        k = np.stack([np.floor(s[:, 0] / 0.5), np.floor(s[:, 1] / 0.5), np.floor(s[:, 2] / (10 * np.pi / 180))])
        return torch.from_numpy(k.astype(np.int64))

    def draw_window(self, rng, m0, m1, sums, sums_are_totals, rank, world, window, flags):
        random = {}
        if self._chain is not None:
            rs = []
            for m in range(m0, m1):
                q, rnd = self._chain[m]
                if rnd:
                    random[m] = lcg_skip(rng, q)
                    rs.append(None)
                else:
                    rs.append(lcg_skip(rng, q + 1) / float(1 << 48))
        else:
            rs = [lcg_skip(rng, 2 * m + 2) / float(1 << 48) for m in range(m0, m1)]
        self._fill_window(rs, m0, sums, sums_are_totals, rank, world, window, flags, random)

    _chain = None
    _n_random = 0
    _w_diff = 0.0

    def _fill_window(self, rs, m0, sums, sums_are_totals, rank, world, window, flags, random=None):
        m1 = m0 + len(rs)
        s = sums.tolist()
        if sums_are_totals:
            T = 0.0
            for v in s:
                T += v
            s = [v / T for v in s]
        offset = 0.0
        for r in range(rank):
            offset += s[r]
        top = offset + s[rank]
        window.zero_()
        w = window.numpy()
        n = self.samples.shape[0]
        cell_th = 10 * np.pi / 180
        for m in range(m0, m1):
            o = m - m0
            if random and m in random:
                if rank == 0:  # the random free-space poses are written by shard 0 only
                    p = self._random_pose(random[m])
                    w[0:3, o] = p.view(np.int64)
                    w[3, o] = int(np.floor(p[0] / 0.5))
                    w[4, o] = int(np.floor(p[1] / 0.5))
                    w[5, o] = int(np.floor(p[2] / cell_th))
                continue
            r = rs[m - m0]
            mine = r >= offset and (r < top or rank == world - 1)
            if not mine:
                continue
            if not r < top:
                flags[0] = 1
                i = n - 1
            elif not r < offset + self.cdf[n]:
                i = n - 1
            else:
                lo, hi = 0, n
                while hi - lo > 1:
                    mid = (lo + hi) // 2
                    if offset + self.cdf[mid] <= r:
                        lo = mid
                    else:
                        hi = mid
                i = lo
            p = self.samples[i, :3]
            o = m - m0
            w[0:3, o] = p.view(np.int64)
            w[3, o] = int(np.floor(p[0] / 0.5))
            w[4, o] = int(np.floor(p[1] / 0.5))
            w[5, o] = int(np.floor(p[2] / cell_th))

    def kld_reset(self):
        self.tree = self.orc.KDTree()

    def kld_feed(self, keys, n, first):
        k = keys.numpy()
        for q in range(n):
            self.tree.insert_key(k[:, q].astype(np.int32), 1.0)
            count = first + q + 1
            if count > self.pfh.resample_limit(self.tree.leaf_count()):
                return count
        return -1

    def kld_feed_window(self, window, n, first):
        return self.kld_feed(window[3:6].contiguous(), n, first)

    kld_device_min = 1 << 62  # the CPU test backend has no device tree: always the ordered host replay

    def kld_stop_window(self, window, n):
        return False, -1, 0, 0

    def kld_counts(self):
        return self.tree.leaf_count(), self.tree.node_count()

    def tail_small(self, x_all, y_all, th_all, m, lo, hi, leaf, bins):
        self.adopt(x_all[lo:hi], y_all[lo:hi], th_all[lo:hi], hi - lo, m, leaf, bins)
        self.converged(x_all, y_all, m)

    def adopt(self, x, y, th, count, global_m, leaf, bins):
        s = np.zeros((count, 4))
        s[:, 0], s[:, 1], s[:, 2] = x.numpy(), y.numpy(), th.numpy()
        s[:, 3] = 1.0 / global_m
        self.samples = s
        self.leaf, self.bins = leaf, bins

    def converged(self, x_all, y_all, m):
        s = np.zeros((m, 4))
        s[:, 0], s[:, 1] = x_all.numpy(), y_all.numpy()
        import ctypes as C
        pct = C.c_float()
        self.conv = self.orc.lib().orc_pf_update_converged(C.byref(self.pfh.pf), s.ctypes.data_as(
            C.POINTER(C.c_double)), m, C.byref(pct))
        self.pct = pct.value

    def update_action(self, odom, data, global_first, global_count):
        """odom = (model, alpha); data = (pose, delta, absolute_motion).  The serial oracle reaches this
        shard's place in the stream by running (and discarding) the draws of the particles before it."""
        model, alpha = odom
        pose, delta, absm = data
        st = self.pfh.pf.rng
        before = np.zeros((global_first, 4))
        st = self.orc.odom_update_action(model, alpha, pose, delta, absm, before, st)
        st = self.orc.odom_update_action(model, alpha, pose, delta, absm, self.samples, st)
        after = np.zeros((global_count - global_first - self.samples.shape[0], 4))
        st = self.orc.odom_update_action(model, alpha, pose, delta, absm, after, st)
        self.pfh.pf.rng = st

    def skip(self, state, n):
        return lcg_skip(state, n)

    def rng_state(self):
        return int(self.pfh.pf.rng)

    def set_rng_state(self, s):
        self.pfh.pf.rng = s

    def state(self):
        class S:
            pass
        st = S()
        st.sample_count = self.samples.shape[0]
        st.converged, st.percent_converged = self.conv, getattr(self, "pct", 0.0)
        st.w_slow, st.w_fast = self.pfh.pf.w_slow, self.pfh.pf.w_fast
        st.total = float(self._total[0])
        return st
