"""GPU box check: torch (bundled HIP runtime) imported BEFORE libslamgpu shares one runtime with it; the other
order leaves torch without a GPU, so bench.py --gpus N imports torch first."""
import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.version.hip)
    import slam_amd as sg
    print("slamgpu devices", sg.device_count())
else:
    import slam_amd as sg
    print("slamgpu devices", sg.device_count())
    import torch
    print("torch avail", torch.cuda.is_available(), torch.version.hip)
import numpy as np
x = torch.arange(8, dtype=torch.float32, device="cuda")
s = sg.SlamGpu(512, 4, rng_mode=sg.RNG_PHILOX, external_stream=torch.cuda.current_stream().cuda_stream)
buf = torch.zeros(2, dtype=torch.float32, device="cuda")
w, nb = s.shard_block_totals()
s.predict(1.0, 0.0, np.eye(2, dtype=np.float32) * 0.01, 0.025)
s.shard_update(np.zeros((0, 2), np.float32), np.zeros(0, np.int32), np.array([[5.0, 0.1]], np.float32), np.eye(2, dtype=np.float32) * 1e-2)
s.dev_copy(buf.data_ptr(), w, 8)
torch.cuda.synchronize()
print("buf", buf.cpu().numpy(), float(x.sum()))
assert abs(float(buf[0]) - 0.5) < 1e-5 and float(x.sum()) == 28.0
print("COEXIST_OK")
os.system("cat /proc/%d/maps | grep -E 'libamdhip64|librccl|libhsa-runtime' | awk '{print $6}' | sort -u" % os.getpid())
