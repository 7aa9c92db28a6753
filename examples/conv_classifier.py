"""Conv2d MNIST classifier on the HIP path -- the module graph of the reference's
examples/convolutional_digits_classifier.ipynb (cell 2): BASELINE config 5, 28x28x1 images, batch 256.
`python examples/conv_classifier.py --steps 50` trains it on synthetic digits (one bright blob per class position)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "numpy-nn-model_amd"))
import neunet_hip  # noqa: E402,F401
import neunet_hip.nn as nn  # noqa: E402


class Conv2dClassifier(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 8, 3, 1, 1)
        self.maxpool1 = nn.MaxPool2d(2, 2)
        self.conv2 = nn.Conv2d(8, 16, 3, 1, 1)
        self.maxpool2 = nn.MaxPool2d(2, 2)
        self.bnorm = nn.BatchNorm2d(16)
        self.leaky_relu = nn.LeakyReLU()
        self.fc1 = nn.Linear(784, 10)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        x = self.conv1(x)
        x = self.leaky_relu(x)
        x = self.maxpool1(x)
        x = self.conv2(x)
        x = self.leaky_relu(x)
        x = self.maxpool2(x)
        x = self.bnorm(x)
        x = x.reshape(x.shape[0], -1)
        x = self.fc1(x)
        return self.sigmoid(x)


def main():
    import argparse

    import numpy as np
    from neunet_hip import Tensor
    from neunet_hip.optim import Adam
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    model = Conv2dClassifier().to("cuda")
    opt = Adam(model.parameters(), lr=2e-3)
    loss_fn = nn.MSELoss()
    for step in range(args.steps):
        y = rng.integers(0, 10, args.batch)
        x = rng.standard_normal((args.batch, 1, 28, 28)).astype(np.float32) * 0.1
        for i, c in enumerate(y):                     # class c = a bright 6x6 blob at a class-specific position
            r0, c0 = 2 + 5 * (c // 5) * 2, 1 + 5 * (c % 5)
            x[i, 0, r0:r0 + 6, c0:c0 + 6] += 1.0
        onehot = np.eye(10, dtype=np.float32)[y]
        out = model(Tensor(x, device="cuda"))
        loss = loss_fn(out, Tensor(onehot, device="cuda", requires_grad=False))
        loss.backward()
        opt.step()
        opt.zero_grad()
        if step % 10 == 0 or step == args.steps - 1:
            acc = float((np.argmax(out.numpy() if hasattr(out, "numpy") else np.asarray(out.data.cpu()), 1) == y).mean())
            print(f"step {step:4d}  loss {loss.item():.4f}  acc {acc:.2f}")


if __name__ == "__main__":
    main()

