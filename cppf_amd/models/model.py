"""Drop-in for the pair-encoder half of the reference's models/model.py.

`ResLayer` and `PPFEncoder` keep the reference's constructor signatures, parameter names and shapes
(models/model.py:8-31, 80-87), so reference checkpoints load with `load_state_dict`
(nocs/inference.py:88), and the same call signatures (models/model.py:89-91, 117):

    PPFEncoder(ppffcs, out_dim)
    .forward(pc[1,N,3], pc_normal[1,N,3], feat[1,N,F], dist=None, idxs=None) -> f32[1,P,out_dim]
    .forward_with_idx(pc[N,3], pc_normal[N,3], feat[N,F], idxs[P,2])        -> f32[P,out_dim]

Under `torch.no_grad()` (inference, nocs/inference.py:179-182) the whole of forward_with_idx -- PPF
construction, gather/concat, three ResLayers and the final linear -- is one HIP kernel
(csrc/pair_mlp.hip).  When autograd needs the graph (train.py:66,91) and the shape is the standard one
(ppffcs [84,32,32,16]) the forward is the same HIP kernel wrapped in an autograd.Function whose backward is
csrc/pair_mlp_bwd.hip (gradients of every parameter and of `feat`); other
shapes evaluate the composite of torch ops.

`PointEncoder` (SPRIN, models/model.py:36-78) produces the per-point `feat` the pair encoder gathers:
    PointEncoder(k, spfcs, out_dim, num_layers=2, num_nbr_feats=2)
    .forward(pc[1,N,3], pc_normal[1,N,3], dist[1,N,N])      -> f32[1,N,out_dim + out_dim//4]
    .forward_nbrs(pc, pc_normal, nbrs_idx[1,N,k])           -> same
Under `no_grad` it is kNN selection + one fused HIP kernel per layer (csrc/sprin.hip); `dist=None` is
accepted as an extension and selects neighbours from exact squared distances without an N x N matrix.
"""
import ctypes as C
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .._torch_util import require_cuda, stream_ptr, workspace
from .sprin import GlobalInfoProp, SparseSO3Conv, pack_point_encoder

__all__ = ["ResLayer", "PPFEncoder", "PointEncoder"]


class ResLayer(nn.Module):
    """models/model.py:8-31: fc1 -> relu -> fc2, plus fc0(x) (or x when dim_in == dim_out)."""

    def __init__(self, dim_in, dim_out, bn=False):
        super().__init__()
        assert bn is False
        self.fc1 = nn.Linear(dim_in, dim_out)
        self.fc2 = nn.Linear(dim_out, dim_out)
        self.fc0 = nn.Linear(dim_in, dim_out) if dim_in != dim_out else None

    def forward(self, x):
        x_res = x if self.fc0 is None else self.fc0(x)
        return self.fc2(F.relu(self.fc1(x))) + x_res


def _params_of(module):
    """The module's parameters as a cached list.  nn.Module.parameters() walks the module tree on every call (~100 us for
    these encoders, several times per training step); the Parameter objects themselves are stable: .to() / .cuda() /
    load_state_dict() swap or fill their data in place, and the encoders' structure is fixed by their constructors.
    Code that REPLACES a Parameter object or a submodule afterwards must call the encoder's invalidate()."""
    plist = module.__dict__.get("_cppf_plist")
    if plist is None:
        plist = list(module.parameters())
        module.__dict__["_cppf_plist"] = plist
    return plist


_STREAM_STATE = weakref.WeakKeyDictionary()   # encoder -> {"readers": {stream: event}, "image_ev": (event, stream, generation)}


class _DeviceWeights:
    """What both encoders share about their device-side weight images.

    The HIP kernels read a lane-ordered IMAGE of the parameters, rebuilt when a parameter changes.  A change is detected
    through (data_ptr, _version) of every Parameter -- what optimizers, load_state_dict(), .to() and any in-place torch
    op on the Parameter update.  Writes that bypass the version counter (`p.data.copy_()`, `p.data.mul_()`, EMA / weight
    surgery through `.data`) are invisible to it: call `invalidate()` after them.  The image is rebuilt IN PLACE (same
    device buffer) whenever size and device allow, so captured hipGraphs that baked its address in stay valid and pick
    the new weights up at their next replay (inference.CenterPipeline re-checks before every replay)."""

    def invalidate(self):
        """Forget every cached view of the parameters (weight images, flat copies, parameter lists)."""
        self.__dict__["_cppf_epoch"] = self.__dict__.get("_cppf_epoch", 0) + 1
        for k in ("_cppf_plist", "_cppf_ordered", "_cppf_transposed"):
            self.__dict__.pop(k, None)

    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .float(): parameters may move or be re-created
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return out

    def _param_key(self, device):
        return (str(device), self.__dict__.get("_cppf_epoch", 0)) + tuple((p.data_ptr(), p._version) for p in _params_of(self))

    # Several HIP streams may share one encoder (inference.CenterPipeline lanes, batch.BatchPoseRunner).  The image is rebuilt
    # in place on whichever stream notices the parameter change, so both directions are ordered with events: a rebuild waits for
    # every replay that was still reading the old image (`_note_image_read`), and a reader on another stream waits for the
    # rebuild before its next replay (`_await_image`).  A training loop on one stream pays one event record per re-pack.
    # (the events live outside the module: copy.deepcopy / pickling of an encoder must not meet a HIP event)
    def _stream_state(self):
        st = _STREAM_STATE.get(self)
        if st is None:
            st = _STREAM_STATE[self] = {"readers": {}, "image_ev": (None, None, 0)}
        return st

    def _image_rebuild_begins(self, dev):
        """called right before the weight image is rewritten in place on the current stream of `dev`"""
        readers = self._stream_state()["readers"]
        if readers:
            cur = torch.cuda.current_stream(dev)
            for sid, ev in readers.items():
                if sid != cur.cuda_stream:
                    cur.wait_event(ev)

    def _image_rebuilt(self, dev):
        """called right after the rebuild was enqueued: later readers on other streams wait for this event"""
        st = self._stream_state()
        # recorded after EVERY pack, also the very first one (no reader registered yet): a pipeline that is captured or replayed on
        # another stream right afterwards finds the key unchanged and would otherwise never learn that it has to wait
        if torch.cuda.is_current_stream_capturing():
            return                                   # (packs happen before a capture begins; an event record would become a graph node)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        st["image_ev"] = (ev, torch.cuda.current_stream(dev).cuda_stream, st["image_ev"][2] + 1)

    def _await_image(self, dev, seen_gen):
        """reader side, before a replay on the current stream: wait for the newest rebuild if this reader has not yet;
        returns the rebuild generation to remember"""
        ev, sid, gen = self._stream_state()["image_ev"]
        if ev is not None and gen != seen_gen:
            cur = torch.cuda.current_stream(dev)
            if cur.cuda_stream != sid:
                cur.wait_event(ev)
        return gen

    def _note_image_read(self, dev):
        """reader side, after a replay was enqueued on the current stream"""
        readers = self._stream_state()["readers"]
        cur = torch.cuda.current_stream(dev)
        ev = readers.get(cur.cuda_stream)
        if ev is None:
            ev = readers[cur.cuda_stream] = torch.cuda.Event()
        ev.record(cur)


class _PairMlpFunction(torch.autograd.Function):
    """forward_with_idx with a HIP backward (train.py:66,91).  Saves only the inputs; the backward kernel
    recomputes the forward.  Gradients: feat, then the parameters in `flatten_state_dict` order."""

    @staticmethod
    def forward(ctx, enc, pc, pc_normal, feat, idxs, *params):
        with torch.no_grad():
            out = enc._forward_device(pc, pc_normal, feat, idxs)
        ctx.enc = enc
        ctx.save_for_backward(pc, pc_normal, feat, idxs, *params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        enc = ctx.enc
        pc, nrm, feat, idxs, *params = ctx.saved_tensors
        dev = pc.device
        flat, offs_c = enc._flat_params(dev)                   # the forward's flat copy unless a parameter changed since
        dims = (C.c_int * len(enc.ppffcs))(*enc.ppffcs)
        L = _lib.lib()
        P, F_ = idxs.shape[0], feat.shape[1]
        grad_out = grad_out.detach().float().contiguous()
        gp = torch.empty_like(flat)
        gf = torch.zeros((feat.shape[0], F_), dtype=torch.float32, device=dev)
        need = L.cppf_pair_mlp_backward_workspace_bytes(P, feat.shape[0], F_, dims, len(enc.ppffcs) - 1, enc.out_dim)
        ws = workspace(max(int(need), 256), dev, "pair_mlp_bwd")
        pcc, nrmc, featc = (pc.detach().float().contiguous(), nrm.detach().float().contiguous(),
                            feat.detach().float().contiguous())
        with torch.cuda.device(dev):
            rc = L.cppf_pair_mlp_backward(pcc.data_ptr(), nrmc.data_ptr(), featc.data_ptr(), idxs.data_ptr(),
                                          1 if idxs.dtype == torch.int64 else 0, flat.data_ptr(), offs_c, pc.shape[0], F_,
                                          dims, len(enc.ppffcs) - 1, P, enc.out_dim, grad_out.data_ptr(), gp.data_ptr(),
                                          gf.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(dev))
        _lib.check(rc, "cppf_pair_mlp_backward")
        grads, pos = [], 0
        for p in params:
            grads.append(gp[pos:pos + p.numel()].reshape(p.shape))
            pos += p.numel()
        return (None, None, None, gf if ctx.needs_input_grad[3] else None, None, *grads)


class _PointEncoderFunction(torch.autograd.Function):
    """PointEncoder.forward under autograd with a HIP backward (train.py:62-64,91): forward = the inference kernels
    (kNN from `dist` + SPRIN), backward = cppf_point_encoder_backward, which recomputes the forward per point.  Points
    and normals carry no gradient (train.py:58-60); gradients flow to the parameters, in `_ordered_params` order."""

    @staticmethod
    def forward(ctx, enc, pc, nrm, nbrs, *params):
        with torch.no_grad():
            # the per-point contraction [rank 32 x n_nbr_feats], kept for the backward
            mixed = torch.empty((pc.shape[0], 32 * enc.num_nbr_feats), dtype=torch.float32, device=pc.device)
            out = enc._forward_device(pc, nrm, nbrs, keep_contraction=mixed)
        ctx.enc = enc
        # the parameters are saved too: the backward recomputes the forward from the weight image of the CURRENT parameters,
        # so an in-place update between forward and backward must fail autograd's version check instead of giving
        # silently wrong gradients
        ctx.save_for_backward(pc, nrm, nbrs, out, mixed, *params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        enc = ctx.enc
        pc, nrm, nbrs, out, mixed = ctx.saved_tensors[:5]
        dev = pc.device
        packed, desc = enc._packed_weights(dev)
        hid = (C.c_int * len(desc["hidden"]))(*desc["hidden"])
        L = _lib.lib()
        N, k = nbrs.shape
        g = grad_out.detach().float().contiguous()
        n_nat = sum(p.numel() for p in enc._ordered_params())
        gp = torch.empty(n_nat, dtype=torch.float32, device=dev)
        ws = workspace(max(int(L.cppf_point_encoder_backward_workspace_bytes(N)), 256), dev, "point_encoder_bwd")
        with torch.cuda.device(dev):
            rc = L.cppf_point_encoder_backward(pc.data_ptr(), nrm.data_ptr(), nbrs.data_ptr(), N, k, packed.data_ptr(), hid,
                                               len(desc["hidden"]), desc["rank"], desc["n_nbr_feats"], desc["n_out"],
                                               desc["n_glob"], enc.num_layers, out.data_ptr(), mixed.data_ptr(), g.data_ptr(),
                                               gp.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(dev))
        _lib.check(rc, "cppf_point_encoder_backward")
        grads, pos = [], 0
        for p, transposed in zip(enc._ordered_params(), enc._ordered_transposed()):
            v = gp[pos:pos + p.numel()]
            grads.append(v.reshape(p.shape[1], p.shape[0]).t() if transposed else v.reshape(p.shape))
            pos += p.numel()
        return (None, None, None, None, *grads)


class PointEncoder(_DeviceWeights, nn.Module):
    """models/model.py:36-78.  Same constructor, parameter names and call signatures as the reference."""

    def __init__(self, k, spfcs, out_dim, num_layers=2, num_nbr_feats=2):
        super().__init__()
        self.k = int(k)
        self.spfcs = [int(h) for h in spfcs]
        self.out_dim = int(out_dim)
        self.num_layers = int(num_layers)
        self.num_nbr_feats = int(num_nbr_feats)
        self.spconvs = nn.ModuleList([SparseSO3Conv(32, num_nbr_feats, out_dim, *spfcs)])
        self.aggrs = nn.ModuleList([GlobalInfoProp(out_dim, out_dim // 4)])
        for _ in range(num_layers - 1):
            self.spconvs.append(SparseSO3Conv(32, out_dim + out_dim // 4, out_dim, *spfcs))
            self.aggrs.append(GlobalInfoProp(out_dim, out_dim // 4))
        self._packed = None
        self._packed_key = None

    # ------------------------------------------------------------------ reference signatures
    def forward(self, pc, pc_normal, dist=None):
        if self._needs_graph(pc):
            if self._has_device_backward(pc, pc_normal):
                pc2, nrm2 = self._check_inputs(pc, pc_normal)
                d2 = None if dist is None else dist.detach().reshape(pc2.shape[0], pc2.shape[0]).float().contiguous()
                out = _PointEncoderFunction.apply(self, pc2, nrm2, self.neighbours(pc2, d2), *self._ordered_params())
                return out.reshape(*pc.shape[:-1], -1)
            if dist is None:
                dist = torch.cdist(pc, pc)
            nbrs = torch.topk(dist, self.k, largest=False, sorted=False)[1]          # models/model.py:47
            return self._composite(pc, pc_normal, nbrs)
        pc2, nrm2 = self._check_inputs(pc, pc_normal)
        if dist is not None:
            if dist.shape[-2:] != (pc2.shape[0], pc2.shape[0]):
                raise ValueError(f"dist must be [..., N, N], got {tuple(dist.shape)}")
            dist = dist.detach().reshape(pc2.shape[0], pc2.shape[0]).float().contiguous()
        return self._forward_device(pc2, nrm2, self.neighbours(pc2, dist)).reshape(*pc.shape[:-1], -1)

    def forward_nbrs(self, pc, pc_normal, nbrs_idx):
        if self._needs_graph(pc):
            if self._has_device_backward(pc, pc_normal):
                pc2, nrm2 = self._check_inputs(pc, pc_normal)
                nbrs = nbrs_idx.reshape(pc2.shape[0], -1).to(device=pc2.device, dtype=torch.int32).contiguous()
                out = _PointEncoderFunction.apply(self, pc2, nrm2, nbrs, *self._ordered_params())
                return out.reshape(*pc.shape[:-1], -1)
            return self._composite(pc, pc_normal, nbrs_idx)
        pc2, nrm2 = self._check_inputs(pc, pc_normal)
        nbrs = nbrs_idx.reshape(pc2.shape[0], -1).to(device=pc2.device, dtype=torch.int32).contiguous()
        return self._forward_device(pc2, nrm2, nbrs).reshape(*pc.shape[:-1], -1)

    # ------------------------------------------------------------------ device path
    def neighbours(self, pc, dist=None):
        """i32[N,k]: rows of torch.topk(dist, k, largest=False) in ascending index order; without `dist`
        the keys are exact squared distances computed from pc."""
        N = pc.shape[0]
        if self.k > N:
            raise ValueError(f"k={self.k} neighbours requested from {N} points")
        nbrs = torch.empty((N, self.k), dtype=torch.int32, device=pc.device)
        with torch.cuda.device(pc.device):
            rc = _lib.lib().cppf_knn(pc.data_ptr(), 0 if dist is None else dist.data_ptr(), N, self.k, nbrs.data_ptr(),
                                     stream_ptr(pc.device))
        _lib.check(rc, "cppf_knn")
        return nbrs

    def _forward_device(self, pc, nrm, nbrs, keep_contraction=None):
        N, k = nbrs.shape
        packed, desc = self._packed_weights(pc.device)
        hid = (C.c_int * len(desc["hidden"]))(*desc["hidden"])
        L = _lib.lib()
        W = desc["n_out"] + desc["n_glob"]
        out = torch.empty((N, W), dtype=torch.float32, device=pc.device)
        ws = workspace(L.cppf_point_encoder_workspace_bytes(N, desc["n_out"], desc["n_glob"], self.num_layers), pc.device,
                       "point_encoder")
        with torch.cuda.device(pc.device):
            if keep_contraction is not None:   # training: the backward reuses the per-point contraction
                rc = L.cppf_point_encoder_forward_train(pc.data_ptr(), nrm.data_ptr(), nbrs.data_ptr(), N, k, packed.data_ptr(),
                                                        hid, len(desc["hidden"]), desc["rank"], desc["n_nbr_feats"],
                                                        desc["n_out"], desc["n_glob"], self.num_layers, out.data_ptr(),
                                                        keep_contraction.data_ptr(), ws.data_ptr(), ws.numel(),
                                                        stream_ptr(pc.device))
            else:
                rc = L.cppf_point_encoder_forward(pc.data_ptr(), nrm.data_ptr(), nbrs.data_ptr(), N, k, packed.data_ptr(), hid,
                                                  len(desc["hidden"]), desc["rank"], desc["n_nbr_feats"], desc["n_out"],
                                                  desc["n_glob"], self.num_layers, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  stream_ptr(pc.device))
        if rc == -3:
            raise _lib.CppfError(f"no device kernel for PointEncoder(k={self.k}, spfcs={self.spfcs}, out_dim={self.out_dim}): "
                                 "csrc/sprin.hip covers spfcs=[32,64,32,32], out_dim=32, k<=64 (train.py:34)")
        _lib.check(rc, "cppf_point_encoder_forward")
        return out

    # ------------------------------------------------------------------ internals
    def _needs_graph(self, pc):
        if not torch.is_grad_enabled():
            return False
        return pc.requires_grad or any(p.requires_grad for p in _params_of(self))

    def _has_device_backward(self, pc, pc_normal):
        """csrc/sprin_bwd.hip covers the one-layer standard encoder (train.py:34) when only the parameters need gradients"""
        return (pc.is_cuda and self.num_layers == 1 and self.spfcs == [32, 64, 32, 32] and self.out_dim == 32
                and self.num_nbr_feats == 2 and self.k <= 64 and not pc.requires_grad and not pc_normal.requires_grad
                and self.spconvs[0].layer_norm is not None)

    def _ordered_params(self):
        """parameters in the packed natural order of `pack_point_encoder` (one layer); cached like `_params_of`"""
        cached = self.__dict__.get("_cppf_ordered")
        if cached is not None:
            return cached
        ker = self.spconvs[0].kernel
        ps = []
        for i in range(0, len(ker) - 1, 3):
            ps += [ker[i].weight, ker[i].bias, ker[i + 1].weight, ker[i + 1].bias]
        ps += [ker[len(ker) - 1].weight, ker[len(ker) - 1].bias]
        sc = self.spconvs[0]
        ps += [sc.outnet.weight, sc.outnet.bias, sc.layer_norm.weight, sc.layer_norm.bias,
               self.aggrs[0].linear.weight, self.aggrs[0].linear.bias]
        self.__dict__["_cppf_ordered"] = ps
        return ps

    def _ordered_transposed(self):
        """which of `_ordered_params` are stored transposed in the packed layout (outnet.weight: [C][n_out])"""
        cached = self.__dict__.get("_cppf_transposed")
        if cached is None:
            cached = [p is self.spconvs[0].outnet.weight for p in self._ordered_params()]
            self.__dict__["_cppf_transposed"] = cached
        return cached

    def _composite(self, pc, pc_normal, nbrs_idx):
        """models/model.py:63-78 as torch ops (autograd path)."""
        gather = lambda t: torch.gather(t.unsqueeze(-3).expand(*t.shape[:-1], *t.shape[-2:]), -2,
                                        nbrs_idx[..., None].expand(*nbrs_idx.shape, t.shape[-1]))
        pc_nbrs = gather(pc)
        norm = torch.norm(pc_nbrs - pc.unsqueeze(-2), dim=-1, keepdim=True)
        cos = torch.sum(gather(pc_normal) * pc_normal.unsqueeze(-2), -1, keepdim=True)
        feat = self.aggrs[0](self.spconvs[0](pc_nbrs, torch.cat([norm, cos], -1), pc))
        for spconv, aggr in zip(self.spconvs[1:], self.aggrs[1:]):
            feat = aggr(spconv(pc_nbrs, gather(feat), pc))
        return feat

    def _check_inputs(self, pc, pc_normal):
        require_cuda()
        if not pc.is_cuda:
            raise _lib.CppfError("PointEncoder inference runs on a HIP device only (no CPU fallback); "
                                 "move the module and its inputs to cuda")
        if pc.shape[-1] != 3 or pc_normal.shape != pc.shape or pc.dim() not in (2, 3) or (pc.dim() == 3 and pc.shape[0] != 1):
            raise ValueError("pc / pc_normal must be [1,N,3] (or [N,3])")
        return (pc.detach().reshape(-1, 3).float().contiguous(), pc_normal.detach().reshape(-1, 3).float().contiguous())

    def _packed_weights(self, device):
        key = self._param_key(device)
        if self._packed is not None and self._packed_key == key:
            return self._packed
        dev = torch.device(device)
        L = _lib.lib()
        old = self._packed[0] if self._packed is not None and self._packed[0].device == dev else None
        if dev.type == "cuda" and self.num_layers == 1 and self.spfcs == [32, 64, 32, 32] and self.out_dim == 32 \
                and self.num_nbr_feats == 2 and self.spconvs[0].layer_norm is not None:
            # standard encoder: natural block by one torch.cat, MFMA image by a device kernel -- no host round trip, so
            # a training loop (weights change every step) stays on the stream
            nat = torch.cat([(p.detach().t() if tr else p.detach()).reshape(-1).float()
                             for p, tr in zip(self._ordered_params(), self._ordered_transposed())]).to(dev).contiguous()
            desc = dict(hidden=list(self.spfcs), rank=32, n_nbr_feats=2, n_out=32, n_glob=self.out_dim // 4, num_layers=1)
            hid = (C.c_int * 4)(*self.spfcs)
            n = int(L.cppf_point_encoder_packed_floats(hid, 4, 32, 2, 32, desc["n_glob"], 1))
            packed = old if old is not None and old.numel() == n else torch.empty(n, dtype=torch.float32, device=dev)
            if packed is old:
                self._image_rebuild_begins(dev)
            with torch.cuda.device(dev):
                rc = L.cppf_point_encoder_pack_device(nat.data_ptr(), hid, 4, 32, 2, 32, desc["n_glob"], 1, packed.data_ptr(),
                                                      stream_ptr(dev))
            _lib.check(rc, "cppf_point_encoder_pack_device")
        else:
            sd = {k: v.detach().float().cpu().numpy() for k, v in self.state_dict().items()}
            natural, desc = pack_point_encoder(sd, self.num_layers)
            hid = (C.c_int * len(desc["hidden"]))(*desc["hidden"])
            n = L.cppf_point_encoder_packed_floats(hid, len(desc["hidden"]), desc["rank"], desc["n_nbr_feats"], desc["n_out"],
                                                   desc["n_glob"], self.num_layers)
            image = np.zeros(max(int(n), natural.size), np.float32)
            _lib.check(L.cppf_point_encoder_pack(natural.ctypes.data, hid, len(desc["hidden"]), desc["rank"],
                                                 desc["n_nbr_feats"], desc["n_out"], desc["n_glob"], self.num_layers,
                                                 image.ctypes.data), "cppf_point_encoder_pack")
            if old is not None and old.numel() == image.size:
                packed = old
                self._image_rebuild_begins(dev)
                packed.copy_(torch.from_numpy(image))
            else:
                packed = torch.from_numpy(image).to(device)
        self._packed = (packed, desc)
        self._packed_key = key
        if dev.type == "cuda":
            self._image_rebuilt(dev)
        return self._packed

    def forward_dyn(self, pc, pc_normal, n_dev, out=None, nbrs=None, nbrs_ready=False):
        """Shape-polymorphic forward for captured chains (cppf_knn_dyn + cppf_point_encoder_forward_dyn): pc / pc_normal are
        capacity-sized f32[n_cap,3] device tensors, `n_dev` a device i32 tensor whose first element is the number of valid
        points (>= k: the caller's duty).  Rows >= n of the result are left untouched.  Returns f32[n_cap, out_dim + out_dim//4].
        nbrs_ready: `nbrs` already holds the k-neighbour sets of the cloud (cppf_knn's output for the same k, e.g. left by
        cppf_frame_cloud_dyn, which fits the normals on them): the search is not repeated."""
        require_cuda()
        n_cap = pc.shape[0]
        packed, desc = self._packed_weights(pc.device)
        hid = (C.c_int * len(desc["hidden"]))(*desc["hidden"])
        L = _lib.lib()
        W = desc["n_out"] + desc["n_glob"]
        if out is None:
            out = torch.zeros((n_cap, W), dtype=torch.float32, device=pc.device)
        if nbrs is None:
            nbrs = torch.zeros((n_cap, self.k), dtype=torch.int32, device=pc.device)
        ws = workspace(L.cppf_point_encoder_workspace_bytes(n_cap, desc["n_out"], desc["n_glob"], self.num_layers), pc.device,
                       "point_encoder")
        with torch.cuda.device(pc.device):
            if not nbrs_ready:
                _lib.check(L.cppf_knn_dyn(pc.data_ptr(), n_cap, n_dev.data_ptr(), self.k, nbrs.data_ptr(), stream_ptr(pc.device)),
                           "cppf_knn_dyn")
            rc = L.cppf_point_encoder_forward_dyn(pc.data_ptr(), pc_normal.data_ptr(), nbrs.data_ptr(), n_cap, n_dev.data_ptr(),
                                                  self.k, packed.data_ptr(), hid, len(desc["hidden"]), desc["rank"],
                                                  desc["n_nbr_feats"], desc["n_out"], desc["n_glob"], self.num_layers,
                                                  out.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(pc.device))
        if rc == -3:
            raise _lib.CppfError(f"no device kernel for PointEncoder(k={self.k}, spfcs={self.spfcs}, out_dim={self.out_dim})")
        _lib.check(rc, "cppf_point_encoder_forward_dyn")
        return out


def point_encoder_forward_batch(members):
    """The point encoders of a chain's members in three launches (cppf_point_encoder_forward_batch) instead of three each.
    members: dicts(encoder, pc, nrm, n_dev, out, nbrs[, nbrs_ready]) as PointEncoder.forward_dyn takes them, 1..8 of them, encoders
    of the one-layer standard form with the same k (instances of different categories carry different weights).  -> [out per
    member], bit-equal to forward_dyn member by member; None when the members do not qualify (the caller then loops forward_dyn)."""
    require_cuda()
    e0 = members[0]["encoder"]
    if not 1 <= len(members) <= 8 or any(m["encoder"].num_layers != 1 or m["encoder"].k != e0.k or m["encoder"].spfcs != e0.spfcs
                                         or m["encoder"].out_dim != e0.out_dim or m["encoder"].num_nbr_feats != e0.num_nbr_feats
                                         for m in members):
        return None
    dev = members[0]["pc"].device
    L = _lib.lib()
    arr = (_lib.PointEncItem * len(members))()
    keep = []
    desc = None
    for i, m in enumerate(members):
        packed, desc = m["encoder"]._packed_weights(dev)
        n_cap = m["pc"].shape[0]
        ws = workspace(L.cppf_point_encoder_workspace_bytes(n_cap, desc["n_out"], desc["n_glob"], 1), dev, f"point_encoder{i}")
        a = arr[i]
        a.pc, a.nrm, a.nbrs, a.out = m["pc"].data_ptr(), m["nrm"].data_ptr(), m["nbrs"].data_ptr(), m["out"].data_ptr()
        a.n_dev = None if m.get("n_dev") is None else m["n_dev"].data_ptr()
        a.packed, a.workspace, a.workspace_bytes = packed.data_ptr(), ws.data_ptr(), ws.numel()
        a.n_cap, a.nbrs_ready = n_cap, 1 if m.get("nbrs_ready") else 0
        keep.append((packed, ws))
    hid = (C.c_int * len(desc["hidden"]))(*desc["hidden"])
    with torch.cuda.device(dev):
        rc = L.cppf_point_encoder_forward_batch(len(members), C.cast(arr, C.c_void_p), e0.k, hid, len(desc["hidden"]), desc["rank"],
                                                desc["n_nbr_feats"], desc["n_out"], desc["n_glob"], 1, stream_ptr(dev))
    if rc == -3:
        return None
    _lib.check(rc, "cppf_point_encoder_forward_batch")
    return [m["out"] for m in members]


class PPFEncoder(_DeviceWeights, nn.Module):
    def __init__(self, ppffcs, out_dim):
        super().__init__()
        self.ppffcs = [int(d) for d in ppffcs]
        self.out_dim = int(out_dim)
        self.res_layers = nn.ModuleList(
            ResLayer(self.ppffcs[i], self.ppffcs[i + 1]) for i in range(len(self.ppffcs) - 1))
        self.final = nn.Linear(self.ppffcs[-1], self.out_dim)
        self._packed = None
        self._packed_key = None
        self._flat = None
        self._flat_key = None

    # ------------------------------------------------------------------ reference signatures
    def forward(self, pc, pc_normal, feat, dist=None, idxs=None):
        if idxs is not None:                                   # models/model.py:90-91
            return self.forward_with_idx(pc[0], pc_normal[0], feat[0], idxs)[None]
        # Dense all-pairs branch (models/model.py:92-115; no caller in the reference): pair (i, j)
        # for every i, j, evaluated through the same sampled-pair path.  `dist` is not needed: the
        # pair kernel recomputes ||pc_i - pc_j|| itself.
        if pc.dim() != 3 or pc.shape[0] != 1:
            raise ValueError("dense forward supports batch size 1 (the reference asserts it, train.py:32)")
        n = pc.shape[1]
        ii, jj = torch.meshgrid(torch.arange(n, device=pc.device), torch.arange(n, device=pc.device), indexing="ij")
        allp = torch.stack([ii.reshape(-1), jj.reshape(-1)], -1)
        return self.forward_with_idx(pc[0], pc_normal[0], feat[0], allp).reshape(1, n, n, self.out_dim)

    def forward_with_idx(self, pc, pc_normal, feat, idxs):
        idxs = self._as_index_tensor(idxs, pc.device)
        if self._needs_graph(feat):
            if self._has_device_backward(pc, feat):
                return _PairMlpFunction.apply(self, pc, pc_normal, feat, idxs, *self._ordered_params())
            return self._composite(pc, pc_normal, feat, idxs)
        return self._forward_device(pc, pc_normal, feat, idxs)

    def _forward_device(self, pc, pc_normal, feat, idxs):
        pc, pc_normal, feat = self._check_inputs(pc, pc_normal, feat)
        P = idxs.shape[0]
        out = torch.empty((P, self.out_dim), dtype=torch.float32, device=pc.device)
        dims = (C.c_int * len(self.ppffcs))(*self.ppffcs)
        ws = self._scratch(pc, feat, dims)
        with torch.cuda.device(pc.device):
            rc = _lib.lib().cppf_pair_mlp_forward(
                pc.data_ptr(), pc_normal.data_ptr(), feat.data_ptr(), idxs.data_ptr(),
                1 if idxs.dtype == torch.int64 else 0, self._packed_weights(pc.device).data_ptr(), pc.shape[0],
                feat.shape[1], dims, len(self.ppffcs) - 1, P, self.out_dim, out.data_ptr(), ws.data_ptr(), ws.numel(),
                stream_ptr(pc.device))
        _lib.check(rc, "cppf_pair_mlp_forward")
        return out

    # ------------------------------------------------------------------ fused decode (this package)
    def forward_decode(self, pc, pc_normal, feat, idxs, u_tr, vote_range, u_rot=None, tr_num_bins=32,
                       rot_num_bins=36):
        """forward_with_idx fused with the decode of nocs/inference.py:185-188 (and :245-256 when
        u_rot is given): returns (outputs f32[P,2] = (mu, nu), heads f32[P,8] or None) without ever
        writing the [P,out_dim] logits.  u_* are uniforms in [0,1) standing in for torch.multinomial
        (negative = arg-max bin)."""
        idxs = self._as_index_tensor(idxs, pc.device)
        pc, pc_normal, feat = self._check_inputs(pc, pc_normal, feat)
        P = idxs.shape[0]
        for nm, u in (("u_tr", u_tr), ("u_rot", u_rot)):
            if u is not None and (u.dtype != torch.float32 or not u.is_contiguous() or tuple(u.shape) != (P, 2)
                                  or u.device != pc.device):
                raise ValueError(f"{nm} must be a contiguous f32[P,2] tensor on {pc.device}")
        outputs = torch.empty((P, 2), dtype=torch.float32, device=pc.device)
        heads = torch.empty((P, 8), dtype=torch.float32, device=pc.device) if u_rot is not None else None
        dims = (C.c_int * len(self.ppffcs))(*self.ppffcs)
        L = _lib.lib()
        ws = self._scratch(pc, feat, dims)
        with torch.cuda.device(pc.device):
            rc = L.cppf_pair_mlp_decode(
                pc.data_ptr(), pc_normal.data_ptr(), feat.data_ptr(), idxs.data_ptr(),
                1 if idxs.dtype == torch.int64 else 0, self._packed_weights(pc.device).data_ptr(), pc.shape[0],
                feat.shape[1], dims, len(self.ppffcs) - 1, P, self.out_dim, tr_num_bins, rot_num_bins,
                float(vote_range[0]), float(vote_range[1]), u_tr.data_ptr(),
                u_rot.data_ptr() if u_rot is not None else None, outputs.data_ptr(),
                heads.data_ptr() if heads is not None else None, ws.data_ptr(), ws.numel(), stream_ptr(pc.device))
            if rc == -3:  # architecture / bin counts outside the fused kernel: logits + decode kernels
                logits = self.forward_with_idx(pc, pc_normal, feat, idxs)
                rc = L.cppf_decode_center(logits.data_ptr(), P, self.out_dim, tr_num_bins, float(vote_range[0]),
                                          float(vote_range[1]), u_tr.data_ptr(), outputs.data_ptr(),
                                          stream_ptr(pc.device))
                _lib.check(rc, "cppf_decode_center")
                if heads is not None:
                    rc = L.cppf_decode_rot(logits.data_ptr(), P, self.out_dim, self.out_dim, tr_num_bins, rot_num_bins,
                                           u_rot.data_ptr(), heads.data_ptr(), stream_ptr(pc.device))
        _lib.check(rc, "cppf_pair_mlp_decode")
        return outputs, heads

    def fused_decode_supported(self, tr_num_bins=32, rot_num_bins=36):
        """True when cppf_pair_mlp_decode / cppf_pair_mlp_decode_sel serve this encoder and these bin counts in one fused
        launch (train.py:35's architecture with config/config.yaml's 32 / 36 bins); other configurations go through the logits
        and the stand-alone decode kernels."""
        return (self.ppffcs == [84, 32, 32, 16] and self.out_dim == 141 and int(tr_num_bins) == 32
                and int(rot_num_bins) == 36)

    def forward_decode_sel(self, pc, pc_normal, feat, idxs, u_rot, sel, n_sel, heads, max_sel=None, tr_num_bins=32,
                           rot_num_bins=36):
        """The second MLP pass of nocs/inference.py:236-256 on the pairs that survived the back-vote: for i < min(n_sel[0],
        max_sel) the pair sel[i] gets its heads row {theta_up, theta_right, aux_up, aux_right, sx, sy, sz, 0} written into
        `heads` f32[P,8] (other rows untouched); u_rot f32[P,2] is indexed by original pair.  Must follow a forward_decode /
        forward_with_idx call on the same (feat, parameters) in the same scratch scope: the per-point table it left is reused.
        sel i32[>=max_sel] and n_sel i32[1] are device tensors (cppf_compact_mask's outputs), so no host sync."""
        idxs = self._as_index_tensor(idxs, pc.device)
        pc, pc_normal, feat = self._check_inputs(pc, pc_normal, feat)
        P = idxs.shape[0]
        max_sel = P if max_sel is None else min(int(max_sel), P)
        dims = (C.c_int * len(self.ppffcs))(*self.ppffcs)
        ws = self._scratch(pc, feat, dims)
        with torch.cuda.device(pc.device):
            rc = _lib.lib().cppf_pair_mlp_decode_sel(
                pc.data_ptr(), pc_normal.data_ptr(), feat.data_ptr(), idxs.data_ptr(), 1 if idxs.dtype == torch.int64 else 0,
                self._packed_weights(pc.device).data_ptr(), pc.shape[0], feat.shape[1], dims, len(self.ppffcs) - 1, P,
                self.out_dim, tr_num_bins, rot_num_bins, u_rot.data_ptr(), sel.data_ptr(), n_sel.data_ptr(), max_sel,
                heads.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(pc.device))
            if rc == -3:
                # architecture / bin counts outside the fused kernel: the logits of the selected pairs + cppf_decode_rot, rows
                # scattered back.  The count is read on the host (one sync; not capturable -- the pose pipelines run such
                # configurations in their full-first form instead, see inference.PosePipeline).
                n = min(int(n_sel.item()), max_sel)
                if n > 0:
                    rows = sel[:n].long()
                    logits = self._forward_device(pc, pc_normal, feat, idxs[rows].contiguous())
                    sub = torch.empty((n, 8), dtype=torch.float32, device=pc.device)
                    rc = _lib.lib().cppf_decode_rot(logits.data_ptr(), n, self.out_dim, self.out_dim, tr_num_bins, rot_num_bins,
                                                    u_rot[rows].contiguous().data_ptr(), sub.data_ptr(), stream_ptr(pc.device))
                    _lib.check(rc, "cppf_decode_rot")
                    heads[rows] = sub
                rc = 0
        _lib.check(rc, "cppf_pair_mlp_decode_sel")
        return heads

    # ------------------------------------------------------------------ internals
    def _scratch(self, pc, feat, dims):
        need = _lib.lib().cppf_pair_mlp_workspace_bytes(pc.shape[0], feat.shape[1], dims, len(self.ppffcs) - 1,
                                                        self.out_dim)
        return workspace(max(int(need), 256), pc.device, "pair_mlp")

    def _needs_graph(self, feat):
        if not torch.is_grad_enabled():
            return False
        return feat.requires_grad or any(p.requires_grad for p in _params_of(self))

    def _has_device_backward(self, pc, feat):
        """csrc/pair_mlp_bwd.hip covers ppffcs = [84,32,32,16] (train.py:35) on a HIP device."""
        return (pc.is_cuda and self.ppffcs == [84, 32, 32, 16] and feat.dim() == 2 and feat.shape[1] == 40
                and self.out_dim <= 144)

    def _param_presence(self):
        pres = []
        for layer in self.res_layers:
            pres += [True, True, True, True, layer.fc0 is not None, layer.fc0 is not None]
        return pres + [True, True]

    def _ordered_params(self):
        """parameters in `flatten_state_dict` order: per layer fc1.w, fc1.b, fc2.w, fc2.b, [fc0.w, fc0.b]; final
        (cached like `_params_of`)"""
        cached = self.__dict__.get("_cppf_ordered")
        if cached is not None:
            return cached
        ps = []
        for layer in self.res_layers:
            ps += [layer.fc1.weight, layer.fc1.bias, layer.fc2.weight, layer.fc2.bias]
            if layer.fc0 is not None:
                ps += [layer.fc0.weight, layer.fc0.bias]
        ps += [self.final.weight, self.final.bias]
        self.__dict__["_cppf_ordered"] = ps
        return ps

    def _composite(self, pc, pc_normal, feat, idxs):
        """models/model.py:118-137 as torch ops (autograd path for shapes without a device backward)."""
        a, b = idxs[:, 0].long(), idxs[:, 1].long()
        xy = pc[a] - pc[b]
        d = torch.norm(xy, dim=-1)
        u = xy / (d[..., None] + 1e-7)
        na, nb = pc_normal[a], pc_normal[b]
        ppf = torch.stack([(na * u).sum(-1), (nb * u).sum(-1), (na * nb).sum(-1), d], -1)
        x = torch.cat([feat[a], feat[b], ppf], -1)
        for layer in self.res_layers:
            x = layer(x)
        return self.final(x)

    @staticmethod
    def _as_index_tensor(idxs, device):
        if isinstance(idxs, np.ndarray):                       # nocs/inference.py:177,182
            if idxs.dtype not in (np.int64, np.int32):
                idxs = idxs.astype(np.int64)
            idxs = torch.from_numpy(np.ascontiguousarray(idxs)).to(device, non_blocking=True)
        if not isinstance(idxs, torch.Tensor):
            raise TypeError(f"idxs: expected numpy array or torch tensor, got {type(idxs).__name__}")
        if idxs.dtype not in (torch.int64, torch.int32):
            raise TypeError(f"idxs: expected int64/int32, got {idxs.dtype}")
        if idxs.dim() != 2 or idxs.shape[1] != 2:
            raise ValueError(f"idxs: expected shape [P,2], got {tuple(idxs.shape)}")
        return idxs.to(device).contiguous()

    def _check_inputs(self, pc, pc_normal, feat):
        require_cuda()
        if not pc.is_cuda:
            raise _lib.CppfError("PPFEncoder inference runs on a HIP device only (no CPU fallback); "
                                 "move the module and its inputs to cuda")
        if pc.dim() != 2 or pc.shape[1] != 3 or pc_normal.shape != pc.shape:
            raise ValueError("pc / pc_normal must be [N,3]")
        if feat.dim() != 2 or feat.shape[0] != pc.shape[0] or 2 * feat.shape[1] + 4 != self.ppffcs[0]:
            raise ValueError(f"feat must be [N,F] with 2F+4 == ppffcs[0] == {self.ppffcs[0]}")
        return (pc.detach().float().contiguous(), pc_normal.detach().float().contiguous(),
                feat.detach().float().contiguous())

    def _flat_params(self, device):
        """(flat device f32 copy of the parameters in `flatten_state_dict` order, host i64 offset table), rebuilt when a
        parameter changes (one torch.cat on the device, no host round trip)."""
        key = self._param_key(device)
        if self._flat is None or self._flat_key != key:
            ps = self._ordered_params()
            dev = torch.device(device)
            parts = [p.detach().reshape(-1).float() for p in ps]
            old = self._flat[0] if self._flat is not None else None
            if old is not None and old.device == dev and parts[0].device == dev and old.numel() == sum(t.numel() for t in parts):
                flat = torch.cat(parts, out=old)             # same buffer: see _DeviceWeights
            else:
                flat = torch.cat(parts).to(device).contiguous()
            offs, pos = [], 0
            it = iter(ps)
            for present in self._param_presence():
                if present:
                    offs.append(pos)
                    pos += next(it).numel()
                else:
                    offs.append(-1)
            self._flat = (flat, (C.c_int64 * len(offs))(*offs))
            self._flat_key = key
        return self._flat

    def _packed_weights(self, device):
        """Lane-ordered weight image for the HIP kernels, rebuilt when a parameter changes.  The standard architecture
        packs on the device (cppf_pair_mlp_pack_device: a training loop changes the weights every step and never
        leaves the stream); other stacks pack on the host."""
        key = self._param_key(device)
        if self._packed is not None and self._packed_key == key:
            return self._packed
        dims = (C.c_int * len(self.ppffcs))(*self.ppffcs)
        L = _lib.lib()
        F_ = (self.ppffcs[0] - 4) // 2
        n_res = len(self.ppffcs) - 1
        n = L.cppf_pair_mlp_packed_floats(F_, dims, n_res, self.out_dim)
        if n == 0:
            raise _lib.CppfError(f"no device kernel for ppffcs={self.ppffcs}, out_dim={self.out_dim} "
                                 "(layers wider than 128 units are unsupported)")
        dev = torch.device(device)
        old = self._packed if self._packed is not None and self._packed.device == dev and self._packed.numel() == n else None
        if dev.type == "cuda" and self.ppffcs == [84, 32, 32, 16] and self.out_dim <= 144:
            flat, offs_c = self._flat_params(dev)
            packed = old if old is not None else torch.empty(n, dtype=torch.float32, device=dev)
            if old is not None:
                self._image_rebuild_begins(dev)
            with torch.cuda.device(dev):
                rc = L.cppf_pair_mlp_pack_device(flat.data_ptr(), offs_c, F_, dims, n_res, self.out_dim, packed.data_ptr(),
                                                 stream_ptr(dev))
            _lib.check(rc, "cppf_pair_mlp_pack_device")
            self._packed = packed
        else:
            sd = {k: v.detach().float().cpu().numpy() for k, v in self.state_dict().items()}
            params, offs = flatten_state_dict(sd, self.ppffcs)
            packed = np.zeros(n, np.float32)
            rc = L.cppf_pair_mlp_pack(params.ctypes.data, offs.ctypes.data, F_, dims, n_res, self.out_dim,
                                      packed.ctypes.data)
            _lib.check(rc, "cppf_pair_mlp_pack")
            if old is not None:
                if dev.type == "cuda":
                    self._image_rebuild_begins(dev)
                old.copy_(torch.from_numpy(packed))
                self._packed = old
            else:
                self._packed = torch.from_numpy(packed).to(device)
        self._packed_key = key
        if dev.type == "cuda":
            self._image_rebuilt(dev)
        return self._packed


def flatten_state_dict(sd, ppffcs):
    """state_dict (numpy values, reference key names) -> (flat f32 params, i64 offset table) in the
    layout cppf_pair_mlp_pack() documents: 6 offsets per res layer, then final.weight/bias."""
    chunks, offs, pos = [], [], 0
    names = []
    for i in range(len(ppffcs) - 1):
        names += [f"res_layers.{i}.{k}" for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc0.weight",
                                                   "fc0.bias")]
    names += ["final.weight", "final.bias"]
    for nme in names:
        if nme not in sd:
            offs.append(-1)
            continue
        a = np.ascontiguousarray(sd[nme], dtype=np.float32).reshape(-1)
        chunks.append(a)
        offs.append(pos)
        pos += a.size
    return np.concatenate(chunks).astype(np.float32), np.asarray(offs, dtype=np.int64)


def batch_plan(n_pairs):
    """geometry of forward_decode_batch for lists of these lengths (cppf_pair_mlp_batch_plan): dict(per_xcd, grid, wg_begin) --
    per_xcd > 0: the XCD-pinned mapping (1, 2, 4 or 8 lists of nearly equal length), 0: contiguous workgroup ranges"""
    n = len(n_pairs)
    arr = (C.c_int64 * n)(*[int(v) for v in n_pairs])
    per_xcd, grid, wb = C.c_int(0), C.c_int(0), (C.c_int * (n + 1))()
    _lib.check(_lib.lib().cppf_pair_mlp_batch_plan(n, arr, C.byref(per_xcd), C.byref(grid), wb), "cppf_pair_mlp_batch_plan")
    return dict(per_xcd=per_xcd.value, grid=grid.value, wg_begin=list(wb))


def forward_decode_batch(items, tr_num_bins=32, rot_num_bins=36, tables_out=None):
    """PPFEncoder.forward_decode for up to 8 pair lists in ONE launch (cppf_pair_mlp_decode_batch): the instances of a frame, each
    with its own cloud, pair list and encoder (the reference keeps one network per category, nocs/inference.py:79-90).  `items`:
    dicts {encoder, pc, pc_normal, feat, idxs, u_tr, vote_range[, u_rot]}; all with u_rot or none.  Returns [(outputs, heads)] in
    order -- the same bits as one forward_decode call per item; what the batch saves is the ~9 us a launch spends before its
    first MFMA (weights -> LDS, the first cold index -> gather chain), paid once instead of once per list."""
    if not 1 <= len(items) <= 8:
        raise ValueError("1 to 8 pair lists per launch")
    enc0 = items[0]["encoder"]
    dev = items[0]["pc"].device
    dims = (C.c_int * len(enc0.ppffcs))(*enc0.ppffcs)
    arr = (_lib.PairMlpItem * len(items))()
    keep, outs = [], []
    for i, it in enumerate(items):
        enc = it["encoder"]
        if enc.ppffcs != enc0.ppffcs or enc.out_dim != enc0.out_dim:
            raise ValueError("the encoders of one launch must share an architecture")
        idxs = enc._as_index_tensor(it["idxs"], dev)
        pc, nrm, feat = enc._check_inputs(it["pc"], it["pc_normal"], it["feat"])
        P = idxs.shape[0]
        u_tr, u_rot = it["u_tr"], it.get("u_rot")
        for nm, u in (("u_tr", u_tr), ("u_rot", u_rot)):
            if u is not None and (u.dtype != torch.float32 or not u.is_contiguous() or tuple(u.shape) != (P, 2) or u.device != dev):
                raise ValueError(f"{nm} must be a contiguous f32[P,2] tensor on {dev}")
        outputs = torch.empty((P, 2), dtype=torch.float32, device=dev)
        heads = torch.empty((P, 8), dtype=torch.float32, device=dev) if u_rot is not None else None
        need = _lib.lib().cppf_pair_mlp_workspace_bytes(pc.shape[0], feat.shape[1], dims, len(enc.ppffcs) - 1, enc.out_dim)
        ws = workspace(max(int(need), 256), dev, f"pair_mlp_batch{i}")       # (each list its own per-point table)
        packed = enc._packed_weights(dev)
        a = arr[i]
        a.pc, a.nrm, a.feat, a.idxs, a.packed = pc.data_ptr(), nrm.data_ptr(), feat.data_ptr(), idxs.data_ptr(), packed.data_ptr()
        a.u_tr, a.u_rot = u_tr.data_ptr(), (u_rot.data_ptr() if u_rot is not None else None)
        a.outputs, a.heads = outputs.data_ptr(), (heads.data_ptr() if heads is not None else None)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.n_points, a.n_pairs = pc.shape[0], P
        a.vr0, a.vr1 = float(it["vote_range"][0]), float(it["vote_range"][1])
        a.idx_is_i64 = 1 if idxs.dtype == torch.int64 else 0
        keep.append((idxs, pc, nrm, feat, ws, packed))
        outs.append((outputs, heads))
    with torch.cuda.device(dev):
        rc = _lib.lib().cppf_pair_mlp_decode_batch(len(items), C.cast(arr, C.c_void_p), items[0]["feat"].shape[1], dims,
                                                   len(enc0.ppffcs) - 1, enc0.out_dim, tr_num_bins, rot_num_bins, stream_ptr(dev))
    _lib.check(rc, "cppf_pair_mlp_decode_batch")
    if tables_out is not None:      # the per-point tables this pass left (a batched second pass reuses them: cppf_pose_tail_batch)
        tables_out[:] = [k[4] for k in keep]
    return outs

