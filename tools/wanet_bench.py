"""Time WanetStep (train_generator_wanet.py's loop body) on one GPU: python tools/wanet_bench.py [dataset] [steps].

dataset: cifar10 (B=128, 32 x 32, PreActResNet18) | celeba (B=128, 64 x 64, ResNet18) | imagenet10 (B=32, 224 x 224,
ResNet18(input_size=224): BASELINE config 5 -- the reference cannot run it)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from combat_amd import nets, step as step_mod  # noqa: E402


class Opt:
    noise_rate, ratio, kernel_size, sigma = 0.08, 0.65, 3, (0.1, 1.0)
    pc, target_label, attack_mode, num_classes = 0.5, 0, "all2one", 10
    L2_weight, clean_model_weight, lr_C, lr_G = 0.02, 0.8, 1e-2, 1e-2
    dataset, post_transform_option, random_crop, random_rotation = "cifar10", "use", 5, 10
    s, grid_rescale, bs = 2, 0.15, 128


def main():
    opt = Opt()
    opt.dataset = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    nets.configure_dataset(opt)
    torch.manual_seed(0)
    np.random.seed(0)
    netc, clean = nets.default_classifier(opt).cuda(), nets.default_classifier(opt).cuda().eval()
    netg = nets.GridGenerator(opt).cuda()
    netf = nets.FrequencyModel(2, 3, opt.input_height).cuda().eval()
    st = step_mod.WanetStep(netc, netg, clean, netf, opt)
    gen = torch.Generator().manual_seed(1234)
    pool = []
    for _ in range(4):
        u8 = torch.randint(0, 256, (opt.bs, 3, opt.input_height, opt.input_height), generator=gen, dtype=torch.uint8)
        pool.append((((u8.float() / 255) - 0.5) / 0.5).cuda())
    tg = [torch.randint(0, opt.num_classes, (opt.bs,), generator=gen) for _ in range(4)]
    for i in range(5):
        st.run(pool[i % 4], tg[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        st.run(pool[i % 4], tg[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m = st.read_metrics()
    print("%s B=%d %dx%d: %.2f ms/step, %.0f img/s  (loss_c %.3f, finite %s)" % (
        opt.dataset, opt.bs, opt.input_height, opt.input_height, dt * 1e3, opt.bs / dt, m["loss_c_sum"] / (steps + 5),
        all(np.isfinite(v) for v in m.values())))


if __name__ == "__main__":
    main()
