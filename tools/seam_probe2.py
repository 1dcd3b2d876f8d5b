# a few hundred calls of the pipelined host-buffer seam on a registered buffer, for rocprofv3 --kernel-trace
# --memory-copy-trace (tools/trace_tail.py prints the last call's timeline); BPF_DEBUG_SEAM=1 prints stage clocks per call
import sys, time, numpy as np, os
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
e.registerHostBuffer(buf)
for ch in (0,):
    e.set_option(12, ch)
    for _ in range(300): sc.applyModelToSampleSet(data, buf, 0)
    for _ in range(3): sc.applyModelToSampleSet(data, buf, 0)
