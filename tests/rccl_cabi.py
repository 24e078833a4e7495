"""Worker of tests/test_entrypoints_gpu.py::test_c_abi_allreduce_world1 (not a test module): combat_comm_* / combat_allreduce --
the RCCL wrapper of the C ABI for hosts that are not PyTorch -- on the one GPU of the box: a one-rank communicator, in-place
sums of an fp32 and a bf16 buffer on a side stream (a one-rank sum is the identity), communicator destroyed.  torch is used
for device memory only."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from combat_amd._lib import lib
    torch.cuda.set_device(0)
    uid = ctypes.create_string_buffer(128)
    assert lib.combat_comm_unique_id(uid) == 0, "combat_comm_unique_id"
    comm = ctypes.c_void_p()
    assert lib.combat_comm_init_rank(ctypes.byref(comm), 1, uid, 0) == 0, "combat_comm_init_rank"
    st = torch.cuda.Stream()
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1 << 20, generator=g).cuda()
    b = torch.randn(1 << 16, generator=g).to(torch.bfloat16).cuda()
    a0, b0 = a.clone(), b.clone()
    torch.cuda.synchronize()
    rc1 = lib.combat_allreduce(a.data_ptr(), a.numel(), 0, comm, st.cuda_stream)
    rc2 = lib.combat_allreduce(b.data_ptr(), b.numel(), 1, comm, st.cuda_stream)
    st.synchronize()
    bad = lib.combat_allreduce(a.data_ptr(), a.numel(), 7, comm, st.cuda_stream)       # unknown dtype: refused
    rc3 = lib.combat_comm_destroy(comm)
    print(json.dumps({"rc": [rc1, rc2, rc3], "refused": bad, "f32_identity": bool(torch.equal(a, a0)),
                      "bf16_identity": bool(torch.equal(b, b0))}))


if __name__ == "__main__":
    main()
