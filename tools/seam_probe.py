# quick probe: chunk counts vs ms per update, registered and not
import sys, time, numpy as np
sys.path.insert(0, '.')
import badger_amcl_amd as bpf
from badger_amcl_amd import synth
e = bpf.Engine(0)
size, beams, n = 2000, 1081, 100000
cells, origin = synth.make_map(size); pose = synth.true_pose(size)
ranges, angles = synth.cast_scan(cells, origin, 0.05, pose, beams, seed=5)
m = bpf.OccupancyMap(e, 0.05); m.setCells(cells); m.setOrigin(origin); m.updateDistancesLUTExact(2.0)
sc = bpf.PlanarScanner(e); sc.init(beams, m); sc.setModelLikelihoodField(0.95, 0.05, 0.2, 2.0)
sc.setMapFactors(*synth.MAP_FACTORS); sc.setPlanarScannerPose(synth.SCANNER_POSE)
data = bpf.PlanarData(ranges, angles, 30.0)
s0 = synth.converged_cloud(n, pose); buf = s0.copy()
def t(reps=30):
    for _ in range(5): sc.applyModelToSampleSet(data, buf, 0)
    t0 = time.perf_counter()
    for _ in range(reps): sc.applyModelToSampleSet(data, buf, 0)
    return (time.perf_counter() - t0) / reps * 1e3
for reg in (False, True):
    if reg: e.registerHostBuffer(buf)
    for ch in (1, 2, 3, 0):
        e.set_option(12, ch)
        print("registered" if reg else "pageable  ", "chunks", ch, "%.4f ms" % t(), e.seam_last_plan(), flush=True)
    e.set_option(12, 0)
pf = bpf.ParticleFilter(e, 100, n, 0.0, 0.0, 85.0)
pf.srand48(1)
def cyc(reps=20):
    tt = 0.0
    for k in range(reps + 3):
        buf[:] = s0
        t0 = time.perf_counter()
        pf.initWithSamples(buf); sc.updateSensor(pf, data); pf.updateResample(); pf.getCurrentSet(out=buf)
        if k >= 3: tt += time.perf_counter() - t0
    return tt / reps * 1e3
print("cycle registered %.4f ms" % cyc(), flush=True)
e.unregisterHostBuffer(buf)
print('cycle pageable %.4f ms' % cyc())
