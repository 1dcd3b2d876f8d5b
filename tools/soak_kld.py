"""Randomised comparison of the two forms of the device-side histogram tree (kernels_kld2.hpp: LDS-sized pieces;
kernels_kld.hpp: one launch pair per level) on the set's own tree (bpf_pf_set_samples -> leaf / bin counts): sets of
8 200 .. 150 000 samples, spread, clustered and mixed, sizes around the block-table edges.
usage (GPU box): python tools/soak_kld.py [sets=200]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import badger_amcl_amd as bpf

n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 200
e = bpf.Engine(0)
rng = np.random.default_rng(2024)
bad = forms = 0
declined = 0
for it in range(n_sets):
    kind = it % 5
    n = int(rng.choice([8192, 8193, 9000, 12000, 20000, 40000, 65536, 100000, 150000]))
    s = np.zeros((n, 4)); s[:, 3] = 1.0 / n
    if kind == 0:      # spread over a map of random size
        ext = rng.uniform(5, 200)
        s[:, 0] = rng.uniform(0, ext, n); s[:, 1] = rng.uniform(0, ext, n); s[:, 2] = rng.uniform(-3.14, 3.14, n)
    elif kind == 1:    # a few clusters of random width
        k = int(rng.integers(1, 30)); c = rng.uniform(0, 100, (k, 2)); w = rng.uniform(0.05, 8.0, k)
        which = rng.integers(0, k, n)
        s[:, 0] = c[which, 0] + rng.normal(0, 1, n) * w[which]; s[:, 1] = c[which, 1] + rng.normal(0, 1, n) * w[which]
        s[:, 2] = rng.normal(0, rng.uniform(0.01, 2.0), n)
    elif kind == 2:    # exactly around the top tree's size: few distinct bins
        m = int(rng.choice([1, 2, 2047, 2048, 2049, 2050, 4095, 4096, 4097]))
        bins = rng.integers(-300, 300, (m, 3)); which = rng.integers(0, m, n)
        s[:, 0] = (bins[which, 0] + 0.5) * 0.5; s[:, 1] = (bins[which, 1] + 0.5) * 0.5
        s[:, 2] = (bins[which, 2] % 36 - 18 + 0.5) * (10 * np.pi / 180)
    elif kind == 3:    # half tight, half spread, shuffled or not
        h = n // 2
        s[:h, 0] = rng.normal(50, 0.3, h); s[:h, 1] = rng.normal(50, 0.3, h); s[:h, 2] = rng.normal(0, 0.1, h)
        s[h:, 0] = rng.uniform(0, 100, n - h); s[h:, 1] = rng.uniform(0, 100, n - h); s[h:, 2] = rng.uniform(-3, 3, n - h)
        if rng.random() < 0.5:
            s[:] = s[rng.permutation(n)]
    else:              # a line first (fills the top tree), then a blob: the pieces must decline or cope
        k = int(rng.integers(100, 5000))
        s[:k, 0] = -500 - 0.5 * np.arange(k); s[:k, 1] = -500
        s[k:, 0] = rng.uniform(0, rng.uniform(2, 100), n - k); s[k:, 1] = rng.uniform(0, 50, n - k)
        s[k:, 2] = rng.uniform(-3, 3, n - k)
    pf = bpf.ParticleFilter(e, 100, n, 0.0, 0.0, 85.0)
    got = {}
    for local in (1, 0):
        e.set_option(bpf.pf.OPT_KLD_LOCAL, local)
        pf.initWithSamples(s)
        st = pf.getState()
        got[local] = (st.leaf_count, st.bin_count)
        if local == 1:
            f = e.kld_last_form()
            forms += f == 2
            declined += f == 1
    e.set_option(bpf.pf.OPT_KLD_LOCAL, 1)
    if got[0] != got[1]:
        bad += 1
        print("MISMATCH set %d kind %d n %d: pieces %r levels %r" % (it, kind, n, got[1], got[0]), flush=True)
print("%d sets, %d through the pieces, %d declined to the level loop, %d mismatches" % (n_sets, forms, declined, bad))
sys.exit(1 if bad else 0)
