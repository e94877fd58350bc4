"""CVAE KL term -- counterpart of /root/reference/src/models/components/loss/misc.py:6-26."""
import torch.nn as nn


class KLDivergence(nn.Module):
    def forward(self, mu, logvar):
        if mu is None:
            return 0
        assert mu.size(0) != 0
        mu = mu.reshape(mu.size(0), -1)
        logvar = logvar.reshape(logvar.size(0), -1)
        klds = -0.5 * (1 + logvar - mu.pow(2) - logvar.exp())
        return klds.sum(1).mean(0, True)[0]  # total KL: sum over latent dims, mean over batch
