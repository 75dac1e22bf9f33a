"""Training driver: host-side mirror of lens/train_model.py for the GPU path.

Same surface as the reference (`LENS_Trainer(args)`, `.add_layer`, `.train_layer`, `.save_model`,
`train_model(model, name)`, the saved state_dict keys), same layer-wise schedule: the feature layer is
trained for `epoch_feat` passes over the reference images in shuffled order, then frozen and the output
layer is trained for `epoch_out` passes with spike forcing.  The per-sample arithmetic (forward, clamp,
STDP, weight clamps, ITP, homeostasis) runs in k_train_layer; the host prepares what is sequential
bookkeeping: the image preprocessing (lens/src/dataset.py, same torch ops), the shuffled sample order
(torch's RandomSampler, as DataLoader(shuffle=True) uses), and the annealed learning-rate schedule.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import blitnet as bn
from ._lib import LensAmdError
from .dataset import CustomImageDataset
from .loggers import model_logger


def anneal_schedule(layer, n_steps, horizon=None):
    """(eta_ip, eta_stdp) in effect at each of the n_steps steps (train_model.py:120-128,150-152,179-181): the
    constructor values at step 0; after every 10th step eta = init * ((T - mod)/T)**2, init_itp = 2*eta_stdp.
    T (`horizon`) is the reference's self.T = reference_places * epochs (train_model.py:143), which exceeds the
    number of steps actually taken when the CSV holds fewer rows than --reference_places."""
    T = n_steps if horizon is None else horizon
    eip, est = layer.eta_ip.clone().float(), layer.eta_stdp.clone().float()
    init_itp, init_stdp = layer.eta_stdp.detach().float() * 2, layer.eta_stdp.detach().float()
    out_ip, out_st = torch.zeros(n_steps), torch.zeros(n_steps)
    for mod in range(n_steps):
        out_ip[mod], out_st[mod] = eip, est
        if mod % 10 == 0:
            pt = pow(float(T - mod) / T, 2)
            eip, est = torch.mul(init_itp, pt), torch.mul(init_stdp, pt)
    return out_ip, out_st, eip, est


def training_inputs(frames, kernel_size):
    """uint8 [S,H,W] -> float32 [S, dims*dims]: what CustomImageDataset(test=False) + ProcessImage(is_train=True)
    feed the trainer (dataset.py:106-112,36-49,14-26): centre pixel of x/255 (one-hot conv) * 255, gamma
    correction towards a mean of 127.5, clip, / 255.  Same torch CPU ops as the reference, per image."""
    k = kernel_size
    c0 = (k // 2 - 1) % k
    out = []
    for img in frames:
        x = (img.unsqueeze(0) / 255)[:, c0::k, c0::k]
        dims = img.shape[0] // k
        x = x[:, :dims, :dims] * 255                           # conv with a one-hot kernel is an exact selection
        x = x.to(torch.float32)
        mean = torch.mean(x)
        try:
            gamma = math.log(0.5 * 255) / math.log(mean)
            x = torch.pow(x, gamma).clip(0, 255)
        except Exception:
            pass
        out.append((x.reshape(1, 1, -1) / 255).reshape(-1))
    return torch.stack(out)


class LENS_Trainer(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        for name in vars(args):
            setattr(self, name, getattr(args, name))
        self.dataset_file = os.path.join(self.data_dir, self.reference + ".csv")
        self.reference_dir = os.path.join(self.data_dir, self.dataset, self.camera, self.reference)
        self.device = model_logger(self)
        if getattr(self, "nocuda", False) or self.device.type != "cuda":
            raise LensAmdError("lens_amd trains on an MI355X only: no ROCm device is usable (--nocuda was given or "
                               "none is present) and there is no CPU fallback")
        self.layer_dict = {}
        self.layer_counter = 0
        self.input = int(args.dims * args.dims)
        self.feature = int(self.input * args.feature_multiplier)
        self.output = int(args.reference_places)
        self.add_layer("feature_layer", dims=[self.input, self.feature], thr_range=[self.thr_l_feat, self.thr_h_feat],
                       fire_rate=[self.fire_l_feat, self.fire_h_feat], ip_rate=self.ip_rate_feat,
                       stdp_rate=self.stdp_rate_feat, p=[self.f_exc, self.f_inh], device=self.device)
        self.add_layer("output_layer", dims=[self.feature, self.output], thr_range=[self.thr_l_out, self.thr_h_out],
                       fire_rate=[self.fire_l_out, self.fire_h_out], ip_rate=self.ip_rate_out,
                       stdp_rate=self.stdp_rate_out, p=[self.o_exc, self.o_inh], spk_force=True, device=self.device)

    def add_layer(self, name, **kwargs):
        if name in self.layer_dict:
            raise ValueError(f"Layer with name {name} already exists.")
        setattr(self, name, bn.SNNLayer(**kwargs))
        self.layer_dict[name] = self.layer_counter
        self.layer_counter += 1

    # ------------------------------------------------------------------------------------
    def train_layer(self, x_table, labels, layer, prev_layers=None, orders=None):
        """x_table float32 [S, input] (host), labels int [S].  Trains `layer` in place for its number of epochs.
        `orders` (optional): explicit [epochs, S] sample order; default = torch RandomSampler per epoch, which
        is what DataLoader(shuffle=True) draws (train_model.py:231-235)."""
        from . import engine as E
        dev = self.device
        epochs = self.args.epoch_feat if not prev_layers else self.args.epoch_out
        S = x_table.shape[0]
        self.T = int(self.reference_places * epochs)
        if orders is None:
            orders = [list(torch.utils.data.RandomSampler(range(S))) for _ in range(epochs)]
        order = torch.tensor(np.asarray(orders).reshape(-1), dtype=torch.int32)
        eta_ip, eta_stdp, last_ip, last_stdp = anneal_schedule(layer, order.numel(), horizon=self.T)

        table = x_table.to(dev)
        n_in = table.shape[1]
        prev_layer = None
        for pname in (prev_layers or []):                      # frozen earlier layers (train_model.py:163-168)
            prev_layer = getattr(self, pname)
            table = E.layer_forward_clamped(table.contiguous(), n_in, prev_layer.w.weight.detach().to(dev).contiguous(),
                                            prev_layer.thr.detach().reshape(-1).to(dev).contiguous())
            n_in = table.shape[1]
        ld = E.train_in_pad(n_in)
        xt = E.pad_table(table, ld)
        force_idx = mpre = None
        if layer.spk_force:
            idx = (torch.as_tensor(labels)[order.long()] / self.filter).to(torch.int32)     # int(idx[0]), train_model.py:161
            force_idx = idx.to(dev)
            mpre = E.pad_table((table.cpu() / prev_layer.fire_rate.reshape(-1)).to(dev), ld)   # blitnet.py:201-205
        w = layer.w.weight.detach().to(dev).contiguous()
        thr = layer.thr.detach().reshape(-1).to(dev).contiguous()
        E.train_layer(w, thr, layer.havconnCombinedExc.to(torch.uint8).to(dev), layer.havconnCombinedInh.to(torch.uint8).to(dev),
                      layer.fire_rate.reshape(-1).to(dev), bool(layer.have_rate), xt, order.to(dev),
                      eta_ip.to(dev), eta_stdp.to(dev), force_idx=force_idx, mpre_table=mpre)
        layer.w.weight.data = w.cpu()
        layer.thr.data = thr.cpu().reshape(1, -1)
        layer.eta_ip, layer.eta_stdp = last_ip, last_stdp
        return layer

    def save_model(self, model_out):
        torch.save(self.state_dict(), model_out)


def train_model(model, model_name):
    """Layer-wise training over the reference traverse, then save ./lens/models/<name> (train_model.py:213-252)."""
    kernel_size = model.roi_dim // model.dims
    ds = CustomImageDataset(annotations_file=model.dataset_file, img_dir=model.reference_dir, kernel_size=kernel_size,
                            skip=model.filter, max_samples=model.reference_places, test=False)
    x_table = training_inputs(ds.frames(), kernel_size)
    labels = torch.tensor([ds[i][1] for i in range(len(ds))])
    model.train()
    trained = []
    for layer_name, _ in sorted(model.layer_dict.items(), key=lambda item: item[1]):
        model.logger.info(f"Training layer: {layer_name}")
        model.train_layer(x_table, labels, getattr(model, layer_name), prev_layers=list(trained))
        trained.append(layer_name)
    model.eval()
    os.makedirs("./lens/models", exist_ok=True)
    model.save_model(os.path.join("./lens/models", model_name))
    return model
