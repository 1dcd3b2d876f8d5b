"""Motion model, CPU side: the oracle's C restatement of
Odom::updateAction (odom.cpp:74-301) against a second, independent restatement in plain Python
(math.* is the same libm), bit for bit, and the stream bookkeeping of PDFGaussian::draw.
PARITY UNPINNED: the reference holds no test or fixture for Odom::updateAction."""
import math

import numpy as np
import pytest

A, C_, MASK = 0x5DEECE66D, 0xB, (1 << 48) - 1


class Rng:
    def __init__(self, state):
        self.s = state

    def drand48(self):
        self.s = (A * self.s + C_) & MASK
        return self.s / float(1 << 48)

    def gauss(self, sigma):  # pdf_gaussian.cpp:77-97
        while True:
            while True:
                r = self.drand48()
                if r != 0.0:
                    break
            x1 = 2.0 * r - 1.0
            while True:
                r = self.drand48()
                if r != 0.0:
                    break
            x2 = 2.0 * r - 1.0
            w = x1 * x1 + x2 * x2
            if not (w > 1.0 or w == 0.0):
                break
        return sigma * x2 * math.sqrt(-2.0 * math.log(w) / w)


def norm_angle(a):
    r = math.fmod(a + math.pi, 2.0 * math.pi)
    return r + math.pi if r <= 0.0 else r - math.pi


def adiff(a, b):
    return norm_angle(a - b)


def py_update_action(model, al, pose, delta, absm, s, rng):
    a1, a2, a3, a4, a5 = al
    old_th = pose[2] - delta[2]
    dt = math.sqrt(delta[0] * delta[0] + delta[1] * delta[1])
    if model in (1, 3):
        dr = delta[2]
        sd = [a3 * (dt * dt) + a1 * (dr * dr), a4 * (dr * dr) + a2 * (dt * dt), a1 * (dr * dr) + a5 * (dt * dt)]
        if model == 3:
            sd = [math.sqrt(v) for v in sd]
        for p in s:
            b = adiff(math.atan2(delta[1], delta[0]), old_th) + p[2]
            cs, sn = math.cos(b), math.sin(b)
            th = dt + rng.gauss(sd[0])
            rh = dr + rng.gauss(sd[1])
            sh = 0 + rng.gauss(sd[2])
            p[0] += (th * cs + sh * sn)
            p[1] += (th * sn - sh * cs)
            p[2] += rh
    elif model in (0, 2):
        rot1 = 0.0 if dt < 0.01 else adiff(math.atan2(delta[1], delta[0]), old_th)
        rot2 = adiff(delta[2], rot1)
        n1 = min(abs(adiff(rot1, 0.0)), abs(adiff(rot1, math.pi)))
        n2 = min(abs(adiff(rot2, 0.0)), abs(adiff(rot2, math.pi)))
        sd = [a1 * n1 * n1 + a2 * dt * dt, a3 * dt * dt + a4 * n1 * n1 + a4 * n2 * n2, a1 * n2 * n2 + a2 * dt * dt]
        if model == 2:
            sd = [math.sqrt(v) for v in sd]
        for p in s:
            r1 = adiff(rot1, rng.gauss(sd[0]))
            th = dt - rng.gauss(sd[1])
            r2 = adiff(rot2, rng.gauss(sd[2]))
            p[0] += th * math.cos(p[2] + r1)
            p[1] += th * math.sin(p[2] + r1)
            p[2] += r1 + r2
    else:
        dr = delta[2]
        at2, as2, ar2 = absm[0] * absm[0], absm[1] * absm[1], absm[2] * absm[2]
        rot_sd = math.sqrt(a1 * ar2 + a2 * at2)
        trans_sd = math.sqrt(a3 * at2 + a4 * ar2)
        strafe_sd = math.sqrt(a4 * ar2 + a5 * as2)
        for p in s:
            h = p[2] + delta[2] / 2
            ch, sh_ = math.cos(h), math.sin(h)
            b = adiff(math.atan2(delta[1], delta[0]), old_th) + p[2]
            cs, sn = math.cos(b), math.sin(b)
            th = rng.gauss(trans_sd)
            st = rng.gauss(strafe_sd)
            rh = rng.gauss(rot_sd)
            p[0] += (dt * cs)
            p[1] += (dt * sn)
            p[2] += dr
            p[0] += (th * ch + st * sh_)
            p[1] += (th * sh_ - st * ch)
            p[2] += rh


CASES = [
    dict(pose=(3.0, -1.0, 0.7), delta=(0.21, -0.08, 0.12), absm=(0.25, 0.09, 0.15)),
    dict(pose=(0.0, 0.0, 3.1), delta=(0.001, 0.002, -0.4), absm=(0.01, 0.0, 0.4)),     # in-place turn: rot1 = 0
    dict(pose=(10.0, 4.0, -2.9), delta=(-0.3, 0.01, 0.02), absm=(0.3, 0.02, 0.02)),    # backwards
]


@pytest.mark.parametrize("model", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("case", [0, 1, 2])
def test_oracle_motion_matches_python_restatement(orc, model, case):
    c = CASES[case]
    rng0 = 0x1234ABCD330E + case
    alpha = (0.2, 0.15, 0.25, 0.1, 0.3)
    base = np.random.default_rng(model * 7 + case).normal(0, 2.0, (300, 4))
    got = np.ascontiguousarray(base.copy())
    st = orc.odom_update_action(model, alpha, c["pose"], c["delta"], c["absm"], got, rng0)
    want = [[float(v) for v in r] for r in base]  # plain Python floats: every operation is one IEEE double op
    r = Rng(rng0)
    py_update_action(model, alpha, c["pose"], c["delta"], c["absm"], want, r)
    assert st == r.s
    want = np.array(want)
    # gcc -O2 folds the cos / sin pair of one argument into a sincos() call (as it does in a Release
    # build of the reference); glibc's sincos differs from its sin / cos by 1 ulp for a few arguments
    assert np.all(np.abs(got[:, :2] - want[:, :2]) <= np.spacing(np.abs(want[:, :2])))
    assert (got[:, :2] != want[:, :2]).sum() <= 3
    assert np.array_equal(got[:, 2:], want[:, 2:])  # heading and weight: exact


def test_gaussian_draw_skips_an_exact_zero(orc):
    """pdf_gaussian.cpp:83-92: r == 0.0 is drawn again.  State 0 is reached from s with a*s + c = 0."""
    inv_a = pow(A, -1, 1 << 48)
    before_zero = (-C_ * inv_a) & MASK
    r = Rng(before_zero)
    assert r.drand48() == 0.0
    # start two steps earlier so the zero is the second uniform of the first attempt
    s0 = ((before_zero - C_) * inv_a) & MASK
    s = np.zeros((5, 4))
    want = [list(v) for v in s]
    st = orc.odom_update_action(3, (0.2,) * 5, (0, 0, 0), (0.1, 0, 0), (0.1, 0, 0), s, s0)
    rr = Rng(s0)
    py_update_action(3, (0.2,) * 5, (0, 0, 0), (0.1, 0, 0), (0.1, 0, 0), want, rr)
    assert st == rr.s and np.array_equal(s, np.array(want))
