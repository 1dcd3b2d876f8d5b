"""Shared scenario builders for the parity tests: the same seeded inputs go to the HIP
engine (through the reference-shaped host mirror) and to the oracle."""
import numpy as np

from badger_amcl_amd import synth


class Scenario:
    def __init__(self, orc, size=200, n=257, beams=61, cloud="converged", max_dist=2.0, seed=3,
                 frac_max=0.03, frac_nan=0.02, scanner_pose=(0.1, -0.05, 0.2), map_factors=synth.MAP_FACTORS,
                 range_max=30.0):
        self.orc = orc
        self.size, self.res = size, 0.05
        self.cells, self.origin = synth.make_map(size, self.res)
        self.omap = orc.OccupancyMap(self.cells, self.res, self.origin)
        self.lut = self.omap.update_distances_lut(max_dist)  # oracle brushfire = the host LUT handed over
        self.max_dist = max_dist
        self.pose = synth.true_pose(size, self.res)
        self.range_max = range_max
        self.ranges, self.angles = synth.cast_scan(self.cells, self.origin, self.res, self.pose, beams,
                                                   range_max=range_max, seed=seed, frac_max=frac_max,
                                                   frac_nan=frac_nan)
        if cloud == "converged":
            self.samples = synth.converged_cloud(n, self.pose, seed=seed + 10)
        elif cloud == "spread":
            self.samples = synth.spread_cloud(n, size, self.res, seed=seed + 11)
        else:  # mixture: half near the truth, half anywhere (incl. off-map and inside walls)
            a = synth.converged_cloud(n - n // 2, self.pose, seed=seed + 10)
            b = synth.spread_cloud(n // 2, size, self.res, seed=seed + 11)
            self.samples = np.ascontiguousarray(np.concatenate([a, b]))
            self.samples[:, 3] = 1.0 / n
        rng = np.random.default_rng(seed + 20)
        self.samples[:, 3] *= rng.uniform(0.5, 1.5, n)  # non-uniform prior weights
        self.scanner_pose = scanner_pose
        self.map_factors = map_factors

    # ---- the HIP side, in the reference's call order
    def gpu_objects(self, engine, max_beams, model="lf", min_samples=100, max_samples=None, seed=42,
                    alpha=(0.0, 0.0), model_kw=None):
        import badger_amcl_amd as bpf
        n = self.samples.shape[0]
        m = bpf.OccupancyMap(engine, self.res)
        m.setCells(self.cells)
        m.setOrigin(self.origin)
        m.setDistancesLUT(self.lut, self.max_dist)
        sc = bpf.PlanarScanner(engine)
        sc.init(max_beams, m)
        self.configure_gpu_model(sc, model, model_kw)
        sc.setMapFactors(*self.map_factors)
        sc.setPlanarScannerPose(self.scanner_pose)
        pf = bpf.ParticleFilter(engine, min_samples, max_samples or n, alpha[0], alpha[1], 85.0)
        pf.srand48(seed)
        pf.initWithSamples(self.samples)
        data = bpf.PlanarData(self.ranges, self.angles, self.range_max)
        return m, sc, pf, data

    def configure_gpu_model(self, sc, model, kw=None):
        kw = kw or {}
        if model == "lf":
            p = dict(synth.LF_DEFAULTS, **kw)
            sc.setModelLikelihoodField(p["z_hit"], p["z_rand"], p["sigma_hit"], self.max_dist)
        elif model == "beam":
            p = dict(synth.BEAM_DEFAULTS, **kw)
            sc.setModelBeam(p["z_hit"], p["z_short"], p["z_max"], p["z_rand"], p["sigma_hit"], p["lambda_short"])
        elif model == "gompertz":
            p = dict(synth.GOMPERTZ_LAUNCH, **kw)
            sc.setModelLikelihoodFieldGompertz(p["z_hit"], p["z_rand"], p["sigma_hit"], self.max_dist,
                                               p["gompertz_a"], p["gompertz_b"], p["gompertz_c"], p["input_shift"],
                                               p["input_scale"], p["output_shift"])
        elif model == "prob":
            p = dict(synth.LF_DEFAULTS, do_beamskip=0, beam_skip_distance=0.5, beam_skip_threshold=0.3,
                     beam_skip_error_threshold=0.9)
            p.update(kw)
            sc.setModelLikelihoodFieldProb(p["z_hit"], p["z_rand"], p["sigma_hit"], self.max_dist, p["do_beamskip"],
                                           p["beam_skip_distance"], p["beam_skip_threshold"],
                                           p["beam_skip_error_threshold"])
        else:
            raise ValueError(model)

    # ---- the oracle side
    def oracle_planar(self, max_beams, model="lf", kw=None):
        orc = self.orc
        kw = dict(kw or {})
        base = dict(scanner_pose=self.scanner_pose, off_map_factor=self.map_factors[0],
                    non_free_space_factor=self.map_factors[1], non_free_space_radius=self.map_factors[2])
        if model == "lf":
            base.update(synth.LF_DEFAULTS)
            mid = orc.MODEL_LF
        elif model == "beam":
            base.update(synth.BEAM_DEFAULTS)
            mid = orc.MODEL_BEAM
        elif model == "gompertz":
            base.update(synth.GOMPERTZ_LAUNCH)
            mid = orc.MODEL_LF_GOMPERTZ
        else:
            base.update(synth.LF_DEFAULTS)
            base.update(do_beamskip=0, beam_skip_distance=0.5, beam_skip_threshold=0.3,
                        beam_skip_error_threshold=0.9)
            mid = orc.MODEL_LF_PROB
        base.update(kw)
        return orc.planar(mid, max_beams, **base)

    def oracle_apply(self, p, samples, set_converged=0, stats=None):
        return self.orc.planar_apply(p, self.omap, samples, self.ranges, self.angles, self.range_max, set_converged,
                                     stats)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    s = np.maximum(np.abs(b), 1e-300)
    return d / s
