"""CPU stand-in for distill.HipBackend, built on the oracle (TEST INFRASTRUCTURE ONLY).

Lets the trainers' host logic -- class sharding, index sampling, optimiser bookkeeping, the
collectives -- run under ``gloo`` on a machine without a GPU.  It is never importable from
the product package."""
import torch

from oracle import ref_cpu as R


class OracleBackend:
    def __init__(self, net_seeds=None):
        self.net_seeds = net_seeds
        self.params = None
        self.device = torch.device("cpu")

    def new_network(self, seed):
        s = self.net_seeds[seed] if self.net_seeds is not None else seed
        return R.init_params(int(s))[:6]

    def set_weights(self, weights):
        self.params = list(weights) + [None, None]

    def embed_pool(self, pool, index):
        with torch.no_grad():
            return R.convnet3d_embed(pool[index], self.params)

    def embed_keep(self, x):
        xv = x.detach().clone().requires_grad_(True)
        f = R.convnet3d_embed(xv, self.params)
        return f.detach(), (xv, f)

    def embed_backward(self, handle, g):
        xv, f = handle
        (dx,) = torch.autograd.grad(f, xv, g)
        return dx

    def dm_loss(self, f_real, f_syn, nclass):
        d = f_real.shape[1]
        fr = f_real.view(nclass, -1, d)
        fs = f_syn.view(nclass, -1, d).detach().clone().requires_grad_(True)
        loss_c = ((fr.mean(1) - fs.mean(1)) ** 2).sum(1)
        (g,) = torch.autograd.grad(loss_c.sum(), fs)
        return loss_c.detach(), g.reshape(f_syn.shape)

    def group_sum(self, x, groups, per, scale):
        return x.view(groups, per, -1).sum(1) * scale

    def sgd(self, x, buf, g, lr, mu, first):
        with torch.no_grad():
            if first:
                buf.copy_(g)
            else:
                buf.mul_(mu).add_(g)
            x.sub_(lr * buf)

    def hallucinate(self, static, dynamic, sidx, didx, w, b):
        self._hal = None
        return R.hallucinator(static[sidx], dynamic[didx], w, b)

    def hallucinate_backward(self, g_out, static, dynamic, sidx, didx, w, need_static, b=None):
        st = static.detach().clone().requires_grad_(True)
        dy = dynamic.detach().clone().requires_grad_(True)
        wv = w.detach().clone().requires_grad_(True)
        bv = torch.zeros(3, requires_grad=True)
        out = R.hallucinator(st[sidx], dy[didx], wv, bv)
        gs = torch.autograd.grad(out, [st, dy, wv, bv], g_out)
        return gs[1], (gs[0] if need_static else None), gs[2], gs[3]


class _OracleNet(torch.nn.Module):
    """The 8 ConvNet3D tensors as a module (parameters() order of the reference), oracle forward."""

    def __init__(self, params):
        super().__init__()
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in params])
        self.dropout = torch.nn.Dropout(0.5)

    def forward(self, x):
        return R.convnet3d_logits(x, list(self.ps), training=self.training and self.dropout.p > 0, p_drop=self.dropout.p)


class OracleGMOps:
    """CPU stand-in for distill.HipGMOps (gradient matching), built on the oracle."""

    def __init__(self, dis_metric="ours"):
        self.dis_metric = dis_metric

    def make_net(self, params, geo, num_classes):
        return _OracleNet(params).train()

    def param_grads(self, net, x, labels, create_graph):
        loss = torch.nn.functional.cross_entropy(net(x), labels)
        return list(torch.autograd.grad(loss, list(net.parameters()), create_graph=create_graph))

    def match_loss(self, gw_syn, gw_real):
        return R.match_loss(gw_syn, gw_real, self.dis_metric)

    sgd = OracleBackend.sgd

    def train_epoch(self, net, images, labels, optimizer, batch_train):
        # one un-shuffled pass (the tests use a single batch), batch-global standardisation as epoch() does
        for i in range(0, images.shape[0], batch_train):
            img = R.standardise_batch(images[i:i + batch_train])
            loss = torch.nn.functional.cross_entropy(net(img), labels[i:i + batch_train])
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()


class OracleMTTOps:
    """CPU stand-in for distill.HipMTTOps: per-step double backward with torch autograd (fp32)."""

    def grads(self, params, x, labels):
        xv = x.detach().clone().requires_grad_(True)
        pv = [p.detach().clone().requires_grad_(True) for p in params]
        ce = torch.nn.functional.cross_entropy(R.convnet3d_logits(xv, pv), labels)
        g = torch.autograd.grad(ce, pv, create_graph=True)
        return [t.detach() for t in g], (xv, pv, g)

    def hvp(self, handle, v):
        xv, pv, g = handle
        s = sum((a * b).sum() for a, b in zip(g, v))
        out = torch.autograd.grad(s, [xv] + pv)
        return out[0], list(out[1:])

    sgd = OracleBackend.sgd
    hallucinate = OracleBackend.hallucinate
    hallucinate_backward = OracleBackend.hallucinate_backward
