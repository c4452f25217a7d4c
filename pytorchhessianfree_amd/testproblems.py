"""Synthetic workloads of the shapes BASELINE.json names (random-init weights,
random data: there is no network for datasets).  The reference builds these from
torchvision / DeepOBS (``examples/example_utils.py:59-109``), neither of which
is installed here, so the topologies are re-declared.

* ``resnet18_mnist``  : torchvision ResNet-18 topology, conv1 1->64 7x7/2 without
  bias, fc 512->10: 11 175 370 trainable parameters in 62 tensors
  (example_utils.py:86-109).
* ``allcnnc_cifar100``: DeepOBS ``cifar100_allcnnc`` topology (dropout is inert in
  eval mode, which the reference example sets): 1 387 108 parameters
  (example_utils.py:59-83).
* ``mwe_mlp`` / ``small_nn``: examples/run_mwe.py:16-20, example_utils.py:23-56.
"""

import torch
from torch import nn


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout)
            )

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class ResNet18(nn.Module):
    def __init__(self, in_channels=1, num_classes=10):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cfg = [(64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1),
               (128, 256, 2), (256, 256, 1), (256, 512, 2), (512, 512, 1)]
        self.layers = nn.Sequential(*[_BasicBlock(a, b, s) for a, b, s in cfg])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():  # torchvision's init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.avgpool(self.layers(x))
        return self.fc(torch.flatten(x, -3))


class _Bottleneck(nn.Module):
    def __init__(self, cin, width, stride):
        super().__init__()
        cout = 4 * width
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout)
            )

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNet50(nn.Module):
    """torchvision ResNet-50 topology: 25 557 032 parameters at 1000 classes."""

    def __init__(self, in_channels=3, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        blocks, cin = [], 64
        for width, count, stride in [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]:
            for i in range(count):
                blocks.append(_Bottleneck(cin, width, stride if i == 0 else 1))
                cin = 4 * width
        self.layers = nn.Sequential(*blocks)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.avgpool(self.layers(x))
        return self.fc(torch.flatten(x, -3))


class AllCNNC(nn.Module):
    def __init__(self, num_classes=100):
        super().__init__()

        def c(i, o, k, s=1, p=0):
            return [nn.Conv2d(i, o, k, s, p), nn.ReLU()]

        self.net = nn.Sequential(
            nn.Dropout(0.2),
            *c(3, 96, 3, 1, 1), *c(96, 96, 3, 1, 1), *c(96, 96, 3, 2, 1),
            nn.Dropout(0.5),
            *c(96, 192, 3, 1, 1), *c(192, 192, 3, 1, 1), *c(192, 192, 3, 2, 1),
            nn.Dropout(0.5),
            *c(192, 192, 3, 1, 0), *c(192, 192, 1), *c(192, num_classes, 1),
            nn.AdaptiveAvgPool2d((1, 1)),
        )

    def forward(self, x):
        return torch.flatten(self.net(x), -3)  # also for one unbatched sample [C,H,W]


def l2_regularized(loss_function, model, l2=5e-4):
    """``loss + 0.5 * l2 * sum ||W||^2`` over the WEIGHTS (no biases): the regularised
    loss the reference's All-CNN-C example trains (examples/example_utils.py:77-81 adds
    DeepOBS' ``get_regularization_loss()``; DeepOBS -- not vendored, PyPI ``deepobs`` --
    sums ``factor * ||p||_2^2`` over the non-bias parameters and halves the total;
    ``cifar100_allcnnc`` defaults to ``l2_reg = 5e-4``).  The GGN ignores the term (it
    does not depend on the network output); the Hessian gains ``l2 * I`` on the weights."""
    weights = [p for name, p in model.named_parameters() if "bias" not in name]

    def regularized(outputs, targets):
        reg = sum((w * w).sum() for w in weights)
        total = loss_function(outputs, targets) + (0.5 * l2) * reg
        # what the curvature engine may assume about this loss (checked numerically on first use):
        # base loss + sum 0.5 * coef * ||w||^2 over the listed tensors
        total._hf_quadratic = ((float(l2), tuple(weights)),)
        return total

    return regularized


def relu_margin(model, inputs):
    """Smallest |x| over the inputs of every ``nn.ReLU`` module in one forward pass of
    ``model`` (run it in float64).  A ReLU whose input sits within fp32 rounding of zero
    (1e-7 here) may get a different sign from two correct fp32 implementations, which changes
    first-order gradients and curvature products by 1e-4-ish of their max-norm; with ~1.3 M
    pre-activations in the ResNet-18 workload this happens on roughly every third random
    batch.  Parity tests against float64 / the CPU path therefore use data seeds for which
    this margin is comfortably above fp32 rounding (scripts/experiments/margin_search.py)."""
    smallest = [float("inf")]

    def hook(_, args):
        smallest[0] = min(smallest[0], float(args[0].detach().abs().min()))

    handles = [m.register_forward_pre_hook(hook) for m in model.modules() if isinstance(m, nn.ReLU)]
    try:
        with torch.no_grad():
            model(inputs)
    finally:
        for h in handles:
            h.remove()
    return smallest[0]


# data seeds of ``resnet18_mnist(batch_size=32, seed=0)`` whose smallest |ReLU input| in float64 is
# 1.0e-6 ... 5.7e-7 (found by scripts/experiments/margin_search.py over seeds 1000-3999; fp32
# forward passes are good to 1e-7 ... 6e-7 there): bench.py's rank r uses entry r
RESNET18_B32_SEPARATED_SEEDS = (1483, 1377, 2116, 1101, 3097, 1322, 1192, 1187)


def freeze_stem_and_layer1(model):
    """``requires_grad = False`` on the stem (conv1, bn1) and layer1 of a ResNet: the optimizer then works "in the
    subspace of trainable parameters" (reference optimizer.py:121-123, utils.py:31-32; its own test problem freezes
    its first layer, tests/test_utils.py:39-43).  Returns the model."""
    if hasattr(model, "layers"):  # (torchvision's layer1 = the blocks in front of the first STRIDED block)
        blocks = list(model.layers)
        n1 = next((i for i, b in enumerate(blocks) if i > 0 and getattr(b, "downsample", None) is not None), len(blocks))
        layer1 = blocks[:n1]
    else:
        layer1 = [model.layer1]
    for mod in (model.conv1, model.bn1, *layer1):
        for p in mod.parameters():
            p.requires_grad_(False)
    return model


def count_trainable(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def resnet18_mnist(batch_size=32, seed=0, device="cpu", data_seed=None):
    """Model in eval mode (BatchNorm then uses running statistics, which makes the
    curvature a plain sum over samples -- required for batch sharding, SURVEY.md
    section 7), inputs U[0,1) [B,1,28,28], integer targets, CE-mean."""
    torch.manual_seed(seed)
    model = ResNet18(1, 10).eval()
    g = torch.Generator().manual_seed(seed if data_seed is None else data_seed)
    inputs = torch.rand(batch_size, 1, 28, 28, generator=g)
    targets = torch.randint(0, 10, (batch_size,), generator=g)
    return model.to(device), (inputs.to(device), targets.to(device)), nn.CrossEntropyLoss()


def resnet18_mnist_mse(batch_size=32, seed=0, device="cpu", data_seed=None):
    """The same model and inputs with the loss of the reference's own examples and tests (``nn.MSELoss``,
    examples/run_mwe.py:19, tests/test_utils.py:47): targets = the one-hot rows of the same integer targets."""
    model, (inputs, targets), _ = resnet18_mnist(batch_size, seed, "cpu", data_seed)
    onehot = torch.nn.functional.one_hot(targets, 10).to(torch.float32)
    return model.to(device), (inputs.to(device), onehot.to(device)), nn.MSELoss()


def resnet50_small_images(batch_size=32, seed=0, device="cpu", data_seed=None, image=64):
    """BASELINE.json configs[4]: a ResNet-50-sized parameter vector (25 557 032);
    eval-mode BN, inputs U[0,1) [B,3,image,image], 1000 classes, CE-mean."""
    torch.manual_seed(seed)
    model = ResNet50(3, 1000).eval()
    g = torch.Generator().manual_seed(seed if data_seed is None else data_seed)
    inputs = torch.rand(batch_size, 3, image, image, generator=g)
    targets = torch.randint(0, 1000, (batch_size,), generator=g)
    return model.to(device), (inputs.to(device), targets.to(device)), nn.CrossEntropyLoss()


def allcnnc_cifar100(batch_size=32, seed=0, device="cpu", data_seed=None):
    torch.manual_seed(seed)
    model = AllCNNC(100).eval()
    g = torch.Generator().manual_seed(seed if data_seed is None else data_seed)
    inputs = torch.rand(batch_size, 3, 32, 32, generator=g)
    targets = torch.randint(0, 100, (batch_size,), generator=g)
    return model.to(device), (inputs.to(device), targets.to(device)), nn.CrossEntropyLoss()


def mwe_mlp(batch_size=16, dim=10, seed=0, device="cpu"):
    torch.manual_seed(seed)
    model = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.ReLU(), nn.Linear(dim, dim))
    inputs, targets = torch.rand(batch_size, dim), torch.rand(batch_size, dim)
    return model.to(device), (inputs.to(device), targets.to(device)), nn.MSELoss()


def small_nn(batch_size=32, seed=0, device="cpu", freeze_layer1=True):
    torch.manual_seed(seed)
    model = nn.Sequential(
        nn.Linear(7, 5), nn.ReLU(), nn.Sequential(nn.Linear(5, 5), nn.ReLU()), nn.Linear(5, 3)
    )
    if freeze_layer1:
        for p in next(model.children()).parameters():
            p.requires_grad = False
    inputs, targets = torch.rand(batch_size, 7), torch.rand(batch_size, 3)
    return model.to(device), (inputs.to(device), targets.to(device)), nn.MSELoss()
