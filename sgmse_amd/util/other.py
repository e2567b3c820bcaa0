"""The two helpers of reference sgmse/util/other.py that sit on the enhancement path."""
import torch
import torch.nn.functional as F


def pad_spec(Y: torch.Tensor, mode: str = "zero_pad") -> torch.Tensor:
    """Right-pad the frame axis of Y [B,1,F,T] to the next multiple of 64 (reference util/other.py:76-90).
    Pure data movement on the tensor's device; raises NotImplementedError for unknown modes like the reference."""
    T = Y.size(3)
    num_pad = 64 - T % 64 if T % 64 != 0 else 0
    if mode == "zero_pad":
        return F.pad(Y, (0, num_pad, 0, 0))
    if mode in ("reflection", "replication"):
        m = "reflect" if mode == "reflection" else "replicate"
        if Y.is_complex():
            return torch.complex(F.pad(Y.real, (0, num_pad, 0, 0), mode=m), F.pad(Y.imag, (0, num_pad, 0, 0), mode=m))
        return F.pad(Y, (0, num_pad, 0, 0), mode=m)
    raise NotImplementedError("This function hasn't been implemented yet.")


def set_torch_cuda_arch_list():
    """Reference util/other.py:126-141 sets TORCH_CUDA_ARCH_LIST for its import-time JIT build of the CUDA op.
    The HIP library is built ahead of time for gfx950, so there is nothing to set; kept for enhancement.py:12-13."""
    if torch.cuda.is_available():
        print(f"sgmse_amd: using the prebuilt gfx950 HIP library on {torch.cuda.get_device_name(0)}")
