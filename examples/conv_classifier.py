"""Conv2d MNIST classifier on the HIP path -- the module graph of the reference's
examples/convolutional_digits_classifier.ipynb (cell 2): BASELINE config 5, 28x28x1 images, batch 256."""
import neunet_hip.nn as nn


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
