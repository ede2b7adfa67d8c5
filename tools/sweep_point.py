"""tools/sweep_point.py <n> <mode> [spp] — ONE point of tools/size_sweep.py (the Cornell box with n x n cells per box face) in ONE
organisation (auto | fused | wavefront), three renders: the command rocprofv3 --kernel-trace --stats is wrapped around to
see which kernels the time of that point goes to."""
import sys
import size_sweep as ss
import mitransient_amd.mi as mi

n, mode = int(sys.argv[1]), sys.argv[2]
ss.SPP = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
scene = mi.load_dict(ss.cornell(n))
print(n, mode, ss.timed(scene, mode))
