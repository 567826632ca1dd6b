"""diagnostic: resident blocks per CU of the update_phi LDS kernel for the bench shapes (ammsb_update_phi_occupancy)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import ops
for K, wg in ((256, 64), (512, 64), (1024, 64), (2048, 64), (4096, 256)):
    ctx = ops.Context(ops.make_params(100000, K, E=1000000, num_node_sample=32))
    b, w = C.c_int(), C.c_int()
    ctx.check(ctx.lib.ammsb_update_phi_occupancy(ctx.handle, wg, C.byref(b), C.byref(w)))
    print("K=%d wg=%d: %d blocks/CU x %d waves = %d waves/CU (lib %s)" % (K, wg, b.value, w.value, b.value * w.value,
                                                                       os.environ.get("AMMSB_HIP_LIB", "default")))
    ctx.close()
