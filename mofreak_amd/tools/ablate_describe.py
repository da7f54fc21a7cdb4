"""Ablation builds of the gather path's describe_kernel (profiles/r05_frame_loop_experiments.txt): copies of kernels.hip in which the
MIP's frame loads (dnomip), the FREAK box reads from the integral (dnofreak) or both (dnone) are left out -- results are wrong by
construction, only the kernel's time is of interest.  usage: python mofreak_amd/tools/ablate_describe.py dnomip dnofreak dnone
-> mofreak_amd/_exp/libvar_<name>.so; time them with MOFREAK_HIP_LIBRARY=... rocprofv3 --kernel-trace --stats -- python3
mofreak_amd/tools/detector_probe.py 128 4 describe"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mofreak_amd.tools import ab_tile
def rep(*pairs):
    def f(t):
        for old, new in pairs:
            assert t.count(old) == 1, old
            t = t.replace(old, new)
        return t
    return f
NOMIP = (("            if (!DUMP) {\n                Sample sm[kMipPasses];", "            if (false) {\n                Sample sm[kMipPasses];"),
         ("            } else {  // all 361 positions of both buffers", "            } else if (DUMP) {  // all 361 positions of both buffers"))
NOFREAK = (("v0 = mean_intensity(integ, a.pitch, kx, ky, lut_scale[lane]);", "v0 = lane;"),
           ("v = mean_intensity(integ, a.pitch, kx, ky, lut_scale[theta * kNbPoints + lane]);", "v = lane ^ theta;"))
V = {"dnomip": rep(*NOMIP), "dnofreak": rep(*NOFREAK), "dnone": rep(*NOMIP, *NOFREAK)}
for name in sys.argv[1:]:
    ab_tile.build(name, [], {"kernels.hip": V[name]})
