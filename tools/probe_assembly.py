import sys, os
import numpy as np
sys.path.insert(0, ".")
import ipc_amd
from ipc_amd import scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
V, F = scene.make_mat(n)
c = ipc_amd.Context(0)
c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
c.opt_init(0.04, False)
c.set_positions(scene.twist_state(scene.jitter(V, F), 0.5))
c.set_pattern()
ms, by = c.bench_assembly(0.04 ** 2, 50)
print(f"probe={os.environ.get('IPCGPU_ASM_PROBE','0')} n={n} nT={F.shape[0]} avg_ms={ms:.4f} algBytes={by/1e6:.1f}MB -> {by/ms/1e6:.1f} GB/s ({by/ms/1e6/8000*100:.2f}% of 8 TB/s)")
