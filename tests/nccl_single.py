"""Worker of tests/test_entrypoints_gpu.py::test_rccl_streams_beside_the_step_on_one_gpu (not a test module).

A fresh process initialises the `nccl` (= RCCL) process group of world size 1 BEFORE any other GPU work, then runs
the alternated step with that group and COMBAT_FORCE_ALLREDUCE=1 (so that every bucketed gradient all-reduce is
really issued: RCCL's own streams and events beside the step's three queues), and the same step without a group.
Prints one JSON line: whether both runs left bit-identical state, and their steady-state ms/step."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["COMBAT_FORCE_ALLREDUCE"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    port = sys.argv[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % port, world_size=1, rank=0,
                            device_id=torch.device("cuda", 0))
    from tests.dp_rehearsal import Opt, build
    from combat_amd import nets, step as step_mod
    b, steps, warm = 128, 20, 6
    gen = torch.Generator().manual_seed(77)
    batches = []
    for _ in range(4):
        x = ((torch.randint(0, 256, (b, 3, 32, 32), generator=gen, dtype=torch.uint8).float() / 255) - 0.5) / 0.5
        batches.append((x.cuda(), torch.randint(0, 10, (b,), generator=gen)))

    def run(pg, steps, warm):
        netc, netg, clean, netf = build(nets)
        st = step_mod.AlternatedStep(netc, netg, clean, netf, Opt(), process_group=pg)
        t0 = time.perf_counter()
        for i in range(warm + steps):
            if i == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            x, t = batches[i % 4]
            st.run(x, t, step_mod.StepRandomness(5, 0.4, 0.7, [None] * 5))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        state = {k: v.detach().double().clone() for m in (netc, netg) for k, v in m.state_dict().items()}
        return state, st.read_metrics(), ms

    def dist_of(a, b):
        worst = 0.0
        for k in a:
            if a[k].numel() > 1:
                worst = max(worst, float((a[k] - b[k]).norm() / max(float(b[k].norm()), 1e-30)))
        return worst

    # ---- parity after ONE step (some weight gradients use fp32 atomics: two runs of either kind differ in the last
    # bits, and twenty steps at lr 1e-2 amplify that; one step bounds what the collective itself may change: nothing)
    a, b0, b1 = run(dist.group.WORLD, 1, 0), run(None, 1, 0), run(None, 1, 0)
    noise, delta = dist_of(b0[0], b1[0]), dist_of(a[0], b0[0])
    counters_equal = all(a[1][k] == b0[1][k] for k in ("clean_correct", "bd_correct", "train_correct", "clean_model_correct"))
    # ---- 20 steps of each, twice, interleaved
    ms_pg = min(run(dist.group.WORLD, 20, 6)[2] for _ in range(1))
    ms_no = min(run(None, 20, 6)[2] for _ in range(1))
    ms_pg = min(ms_pg, run(dist.group.WORLD, 20, 6)[2])
    long_run = run(None, 20, 6)
    ms_no = min(ms_no, long_run[2])
    print(json.dumps({"one_step_delta": delta, "one_step_noise": noise, "counters_equal": bool(counters_equal),
                      "ms_with_rccl": ms_pg, "ms_without": ms_no, "backend": dist.get_backend(),
                      "finite": all(bool(torch.isfinite(v).all()) for v in long_run[0].values())}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
