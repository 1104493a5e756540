#!/usr/bin/env python3
"""Throughput of the fused CIFAR-10 ConvNet evaluator (gpt_cnn_eval) on synthetic nets / images.
Usage: python tools/cnn_bench.py [--nets 1280] [--images 10000]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpt_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nets", type=int, default=1280)
    ap.add_argument("--images", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    nets = torch.randn(a.nets, 5370, device="cuda", generator=g) * 0.3
    imgs = torch.randn(a.images, 3, 32, 32, device="cuda", generator=g)
    labels = torch.randint(0, 10, (a.images,), device="cuda", generator=g)
    gpt_amd.cnn_test_losses(nets[:8], imgs[:64], labels[:64])
    torch.cuda.synchronize()
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loss, err = gpt_amd.cnn_test_losses(nets, imgs, labels)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flops = a.nets * a.images * 2.0 * (256 * 16 * 27 + 256 * 32 * 144 + 320)
        print(f"{a.nets} nets x {a.images} images: {ms:.1f} ms  {flops / ms / 1e9:.1f} TFLOP/s fp32  "
              f"{a.nets * a.images / ms / 1e3:.2f} M net-images/s  loss[0]={float(loss[0]):.4f} err[0]={float(err[0]):.2f}")


if __name__ == "__main__":
    main()
