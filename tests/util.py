import numpy as np
import torch


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def bf16_round(a):
    """fp32 numpy -> values representable in bf16 (RNE), still fp32."""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def to_dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype=dtype)


def load_sd(golden_dir, name):
    import os
    z = np.load(os.path.join(golden_dir, name))
    return z, {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
