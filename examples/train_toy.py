"""A student DAU network learns to reproduce a fixed teacher DAU layer: the drop-in Python surface (DAUConv2d with the
reference's constructor arguments) inside an ordinary PyTorch training loop.  Offsets (mu1, mu2) are learned through the
operator's own gradients; `mu_learning_rate_factor` scales them inside the op as in the reference
(plugins/tensorflow/src/dau_conv_grad_op.cpp:297-303).

    python examples/train_toy.py [--steps 60]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dau-convnet_amd"))

import torch
import dau_conv


def run(steps=60, seed=0, verbose=True):
    torch.manual_seed(seed)
    dev = torch.device("cuda", 0)
    teacher = dau_conv.DAUConv2d(filters=16, dau_units=(2, 2), max_kernel_size=9, use_bias=False, in_channels=8,
                                 mu1_initializer=dau_conv.random_uniform_initializer(-3, 3),
                                 mu2_initializer=dau_conv.random_uniform_initializer(-3, 3)).to(dev)
    for p in teacher.parameters():
        p.requires_grad_(False)
    student = torch.nn.Sequential(
        dau_conv.DAUConv2d(filters=16, dau_units=(2, 2), max_kernel_size=9, in_channels=8, activation=torch.relu,
                           mu_learning_rate_factor=10),
        dau_conv.DAUConv2d(filters=16, dau_units=(2, 2), max_kernel_size=9, in_channels=16, use_bias=False,
                           mu_learning_rate_factor=10),
    ).to(dev)
    opt = torch.optim.Adam(student.parameters(), lr=3e-3)
    losses = []
    for step in range(steps):
        x = torch.rand(16, 8, 32, 32, device=dev)
        loss = torch.nn.functional.mse_loss(student(x), teacher(x))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if verbose and step % 10 == 0:
            print("step %3d  loss %.5f" % (step, losses[-1]))
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    l = run(ap.parse_args().steps)
    print("loss %.5f -> %.5f" % (l[0], l[-1]))
