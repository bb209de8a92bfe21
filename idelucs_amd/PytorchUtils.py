"""idelucs_amd.PytorchUtils -- the two CLI-reachable encoders (reference idelucs/PytorchUtils.py:6-56).

Parameter names and shapes match the reference modules, so a reference `state_dict` loads as is.
The dense layers run on PyTorch-ROCm (hipBLASLt -> MFMA); `forward(x) -> (softmax_out, latent)`.
"""
import torch.nn as nn


class myNet(nn.Module):
    """model_size='small' (reference PytorchUtils.py:6-31): F'->400->128, heads 128->64 and 128->C."""

    def __init__(self, n_input, n_output):
        super().__init__()
        self.n_input = n_input
        self.layers = nn.Sequential(nn.Linear(n_input, 400), nn.ReLU(), nn.Dropout(p=0.5),
                                    nn.Linear(400, 128), nn.LeakyReLU())
        self.instance = nn.Linear(128, 64)
        self.classifier = nn.Sequential(nn.Dropout(p=0.5), nn.Linear(128, n_output), nn.Softmax(dim=1))

    def forward(self, x):
        x = self.layers(x.view(-1, self.n_input))
        return self.classifier(x), self.instance(x)


class NetLinear(nn.Module):
    """model_size='linear', the default (reference PytorchUtils.py:33-56): F->512->64 (=latent)->C."""

    def __init__(self, n_input, n_output):
        super().__init__()
        self.n_input = n_input
        self.layers = nn.Sequential(nn.Linear(n_input, 512), nn.ReLU(), nn.Dropout(p=0.5), nn.Linear(512, 64))
        self.classifier = nn.Sequential(nn.ReLU(), nn.Dropout(p=0.5), nn.Linear(64, n_output), nn.Softmax(dim=1))

    def forward(self, x):
        latent = self.layers(x.view(-1, self.n_input))
        return self.classifier(latent), latent
