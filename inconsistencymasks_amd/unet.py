"""Host-side mirror of the reference's model builder (unet.py:46-67) on top of libimk.so.

`get_unet(i_height, i_width, i_channels, num_outputmasks, alpha, actifu, actifuout, ...)` keeps the
reference signature and returns a `UNet` whose `.predict(x)` takes uint8/float NHWC batches and returns
float32 probabilities like `tf.keras.Model.predict` does at functions.py:3157/3184/3224.  Weights live in
one flat fp32 device tensor (layout in include/imk.h); torch is used for device memory only.
"""
import ctypes
import math

import numpy as np
import torch

from ._lib import LayerInfo, UnetCfg, check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Plan:
    """RAII wrapper of imk_unet_plan."""

    def __init__(self, h, w, c_in, n_out, alpha, act_out):
        if actifu_code(act_out) is None:
            raise ValueError(f"unsupported output activation {act_out!r} (sigmoid | softmax)")
        ch = [int(v * alpha) for v in (16, 32, 64, 128, 256)]          # unet.py:49-56
        self.cfg = UnetCfg(h, w, c_in, n_out, (ctypes.c_int * 5)(*ch), actifu_code(act_out))
        self.h, self.w, self.c_in, self.n_out, self.alpha, self.act_out = h, w, c_in, n_out, alpha, act_out
        self._p = ctypes.c_void_p()
        check(lib.imk_unet_plan_create(ctypes.byref(self.cfg), ctypes.byref(self._p)), "imk_unet_plan_create")
        self._describe()

    _ws_fn = "imk_unet_workspace_bytes"

    def _describe(self):
        """parameter layout of the created plan (shared with EvalNet plans, which are the same C object)"""
        t, tr = ctypes.c_int64(), ctypes.c_int64()
        check(lib.imk_unet_param_count(self._p, ctypes.byref(t), ctypes.byref(tr)), "imk_unet_param_count")
        self.n_total, self.n_trainable = t.value, tr.value
        self.layers = []
        for i in range(lib.imk_unet_num_layers(self._p)):
            li = LayerInfo()
            check(lib.imk_unet_layer_info(self._p, i, ctypes.byref(li)), "imk_unet_layer_info")
            self.layers.append(dict(name=li.name.decode(), kind=li.kind, ksize=li.ksize, cin=li.cin, cout=li.cout,
                                    off_w=li.off_w, off_b=li.off_b, off_mean=li.off_mean, off_var=li.off_var))
        self.packed_bytes = lib.imk_unet_packed_bytes(self._p)
        self.state_bytes = lib.imk_unet_state_bytes(self._p)

    @property
    def ptr(self):
        return self._p

    def workspace_bytes(self, batch, mode):
        n = getattr(lib, self._ws_fn)(self._p, batch, mode)
        if n < 0:
            check(int(n), self._ws_fn)
        return n

    def __del__(self):
        try:
            if self._p:
                lib.imk_unet_plan_destroy(self._p)
                self._p = None
        except Exception:
            pass


def actifu_code(name):
    return {"sigmoid": 0, "softmax": 1}.get(name)


def he_normal(shape, fan_in, gen):
    """Keras he_normal (unet.py:46 default): truncated normal, stddev sqrt(2/fan_in)/0.87962566."""
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    w = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return w


class UNet:
    """The tiny U-Net of unet.py as a flat parameter vector + a kernel plan."""

    N_STATS = 4     # loss, overflow flag, loss scale, step

    def __init__(self, h, w, c_in, n_out, alpha, act_out="sigmoid", seed=None, device="cuda"):
        self._init_from_plan(Plan(h, w, c_in, n_out, alpha, act_out), seed, device)

    def _init_from_plan(self, plan, seed, device, dense=()):
        self.plan = plan
        self.device = torch.device(device)
        gen = torch.Generator()
        if seed is None:
            # Keras draws a fresh initialisation per get_unet() call (unet.py:46 he_normal, unseeded): a default
            # torch.Generator() would start from the same constant seed every time and every "candidate" would be the
            # same model.  Draw the seed from torch's global stream, so torch.manual_seed() still makes a run repeatable.
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            # Data parallel (one process per GPU): every rank must start from the SAME weights -- only gradients are
            # all-reduced afterwards -- but each rank's global stream is its own.  Rank 0's draw wins.
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                box = [seed]
                dist.broadcast_object_list(box, src=0)
                seed = int(box[0])
        gen.manual_seed(int(seed))
        self.seed = int(seed)
        flat = torch.zeros(self.plan.n_total, dtype=torch.float32)
        for l in self.plan.layers:
            if l["kind"] == 0 and l["name"] in dense:      # Keras Dense default: glorot_uniform
                ci, co = l["cin"], l["cout"]
                lim = math.sqrt(6.0 / (ci + co))
                flat[l["off_w"]:l["off_w"] + ci * co] = (torch.rand(ci * co, generator=gen) * 2 - 1) * lim
            elif l["kind"] == 0:
                k, ci, co = l["ksize"], l["cin"], l["cout"]
                flat[l["off_w"]:l["off_w"] + k * k * ci * co] = he_normal((k * k * ci * co,), k * k * ci, gen)
            else:
                c = l["cout"]
                flat[l["off_w"]:l["off_w"] + c] = 1.0      # gamma
                flat[l["off_var"]:l["off_var"] + c] = 1.0  # moving variance
        self.params = flat.to(self.device)
        self.packed = torch.empty(self.plan.packed_bytes, dtype=torch.uint8, device=self.device)
        self._ws = {}
        self._packed_ok = False      # packed conv weights match params
        self._fold_ok = False        # folded inference BN statistics match params (moving statistics)
        self.train_state = None

    def debug(self, materialize=None, single_stream=None):
        """Debug / measurement switches of this model's plan (imk_unet_plan_debug): materialize -- inference also stores the
        intermediates fused kernels keep on chip; single_stream -- no side streams (exclusive kernel timings)."""
        check(lib.imk_unet_plan_debug(self.plan.ptr, -1 if materialize is None else int(bool(materialize)),
                                      -1 if single_stream is None else int(bool(single_stream))), "imk_unet_plan_debug")

    def set_bn_momentum(self, momentum):
        """momentum of the BatchNorm moving statistics in training steps (Keras default 0.99)"""
        check(lib.imk_unet_plan_set_bn_momentum(self.plan.ptr, float(momentum)), "imk_unet_plan_set_bn_momentum")

    def get_bn_momentum(self):
        """the momentum the plan's training steps use (read back from the library)"""
        m = ctypes.c_float()
        check(lib.imk_unet_plan_get_bn_momentum(self.plan.ptr, ctypes.byref(m)), "imk_unet_plan_get_bn_momentum")
        return float(m.value)

    # ---- parameters -------------------------------------------------------------------------------
    def ready_for_inference(self):
        """training steps re-pack the conv weights but leave the folded BN statistics stale: refresh if needed"""
        if not (self._packed_ok and self._fold_ok):
            self.repack()

    def repack(self):
        check(lib.imk_unet_pack_weights(self.plan.ptr, self.params.data_ptr(), self.packed.data_ptr(), _stream()),
              "imk_unet_pack_weights")
        self._packed_ok = self._fold_ok = True

    def set_params(self, flat):
        self.params.copy_(torch.as_tensor(flat, dtype=torch.float32).to(self.device))
        self._packed_ok = self._fold_ok = False

    def state_dict(self):
        """name -> CPU tensor, Keras shapes (conv kernels HWIO)."""
        out = {}
        p = self.params.detach().cpu()
        for l in self.plan.layers:
            n = l["name"]
            if l["kind"] == 0:
                k, ci, co = l["ksize"], l["cin"], l["cout"]
                out[n + ".w"] = p[l["off_w"]:l["off_w"] + k * k * ci * co].reshape(k, k, ci, co).clone()
                out[n + ".b"] = p[l["off_b"]:l["off_b"] + co].clone()
            else:
                c = l["cout"]
                out[n + ".gamma"] = p[l["off_w"]:l["off_w"] + c].clone()
                out[n + ".beta"] = p[l["off_b"]:l["off_b"] + c].clone()
                out[n + ".mean"] = p[l["off_mean"]:l["off_mean"] + c].clone()
                out[n + ".var"] = p[l["off_var"]:l["off_var"] + c].clone()
        return out

    def load_state_dict(self, sd):
        flat = torch.zeros(self.plan.n_total, dtype=torch.float32)
        for l in self.plan.layers:
            n = l["name"]
            if l["kind"] == 0:
                k, ci, co = l["ksize"], l["cin"], l["cout"]
                flat[l["off_w"]:l["off_w"] + k * k * ci * co] = torch.as_tensor(sd[n + ".w"]).reshape(-1)
                flat[l["off_b"]:l["off_b"] + co] = torch.as_tensor(sd[n + ".b"])
            else:
                c = l["cout"]
                flat[l["off_w"]:l["off_w"] + c] = torch.as_tensor(sd[n + ".gamma"])
                flat[l["off_b"]:l["off_b"] + c] = torch.as_tensor(sd[n + ".beta"])
                flat[l["off_mean"]:l["off_mean"] + c] = torch.as_tensor(sd[n + ".mean"])
                flat[l["off_var"]:l["off_var"] + c] = torch.as_tensor(sd[n + ".var"])
        self.set_params(flat)

    def count_params(self):
        return self.plan.n_total

    # ---- inference ----------------------------------------------------------------------------------
    def workspace(self, batch, mode, extra=0):
        key = (batch, mode, extra)
        if key not in self._ws:
            self._ws = {k: v for k, v in self._ws.items() if k[1] != mode}  # keep one per mode
            n = self.plan.workspace_bytes(batch, mode) + extra
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._ws[key]

    def _as_u8_batch(self, x):
        if isinstance(x, (list, tuple)):       # Keras accepts model.predict([batch])
            x = x[0]
        t = torch.as_tensor(x)
        if t.dim() == 3:
            t = t[None]
        if t.dtype != torch.uint8:
            t = t.round().clamp(0, 255).to(torch.uint8)   # images are 0..255 in every reference call site
        if t.shape[1:] != (self.plan.h, self.plan.w, self.plan.c_in):
            raise ValueError(f"expected [B,{self.plan.h},{self.plan.w},{self.plan.c_in}], got {tuple(t.shape)}")
        return t.to(self.device).contiguous()

    def predict_device(self, x_u8):
        """uint8 device tensor [B,H,W,C] -> float32 device tensor [B,H,W,K]."""
        self.ready_for_inference()
        b = x_u8.shape[0]
        ws = self.workspace(b, 0)
        probs = torch.empty((b, self.plan.h, self.plan.w, self.plan.n_out), dtype=torch.float32, device=self.device)
        check(lib.imk_unet_forward(self.plan.ptr, self.params.data_ptr(), self.packed.data_ptr(), x_u8.data_ptr(), b,
                                   probs.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "imk_unet_forward")
        return probs

    def predict(self, x, batch_size=64, verbose=0):
        """Keras-like: numpy in, numpy float32 [B,H,W,K] out."""
        xt = self._as_u8_batch(x)
        outs = []
        for i in range(0, xt.shape[0], batch_size):
            outs.append(self.predict_device(xt[i:i + batch_size]))
        return torch.cat(outs, 0).cpu().numpy()

    def intermediate(self, layer_name, batch, mode=0, which=0):
        """Debug/parity: a stored intermediate after the last forward (fp16 -> float32 CPU [B,h,w,c])."""
        idx = [l["name"] for l in self.plan.layers].index(layer_name)
        off, h, w, c, cs = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib.imk_unet_tensor_info(self.plan.ptr, batch, mode, idx, which, ctypes.byref(off), ctypes.byref(h),
                                       ctypes.byref(w), ctypes.byref(c), ctypes.byref(cs)), "imk_unet_tensor_info")
        ws = [v for k, v in self._ws.items() if k[0] == batch and k[1] == mode][0]
        n = batch * h.value * w.value * cs.value
        t = ws[off.value:off.value + 2 * n].view(torch.float16).reshape(batch, h.value, w.value, cs.value)
        return t[..., :c.value].float().cpu()

    # ---- training ---------------------------------------------------------------------------------------
    def init_train_state(self):
        self.train_state = torch.empty(self.plan.state_bytes, dtype=torch.uint8, device=self.device)
        check(lib.imk_unet_state_init(self.plan.ptr, self.train_state.data_ptr(), _stream()), "imk_unet_state_init")
        # gradients and the step statistics share one buffer: a data-parallel step is ONE all-reduce (grads_and_stats)
        self.grads_and_stats = torch.zeros(self.plan.n_trainable + self.N_STATS, dtype=torch.float32, device=self.device)
        self.grads = self.grads_and_stats[:self.plan.n_trainable]
        self.stats = self.grads_and_stats[self.plan.n_trainable:]

    def fwd_bwd(self, x_u8, y_u8, loss_kind):
        """One forward/backward on device batches; fills self.grads (unscaled) and self.stats."""
        if self.train_state is None:
            self.init_train_state()
        if not self._packed_ok:
            self.repack()
        self._fold_ok = False        # the step updates the moving statistics
        b = x_u8.shape[0]
        ws = self.workspace(b, 1)
        check(lib.imk_unet_fwd_bwd(self.plan.ptr, self.params.data_ptr(), self.packed.data_ptr(),
                                   self.train_state.data_ptr(), x_u8.data_ptr(), y_u8.data_ptr(), b, int(loss_kind),
                                   self.grads.data_ptr(), self.stats.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
              "imk_unet_fwd_bwd")

    def adamw_step(self, lr, wd, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-7):
        check(lib.imk_unet_adamw_step(self.plan.ptr, self.params.data_ptr(), self.packed.data_ptr(),
                                      self.train_state.data_ptr(), self.grads.data_ptr(), self.stats.data_ptr(),
                                      float(grad_scale), float(lr), float(wd), float(beta1), float(beta2), float(eps),
                                      _stream()), "imk_unet_adamw_step")
        self._packed_ok = True

    def train_step(self, x_u8, y_u8, loss_kind, lr, wd):
        self.fwd_bwd(x_u8, y_u8, loss_kind)
        self.adamw_step(lr, wd)


def get_unet(i_height, i_width, i_channels, num_outputmasks, alpha, actifu, actifuout, ks=3, kernel_ini="he_normal",
             dropout_rate_encoder=0, dropout_rate_decoder=0, dropout_rate_bottleneck=0, seed=None, device="cuda"):
    """Same positional signature as the reference's unet.get_unet (unet.py:46).  Only what every shipped
    config uses is supported by the kernels: relu hidden activation, 3x3 kernels, he_normal, no dropout.

    Data parallel: with `seed=None` and an initialised torch.distributed process group of more than one rank this call is a
    COLLECTIVE (rank 0 draws the seed and broadcasts it, so the replicas start equal): every rank must make it, in the same
    order.  Pass a seed to build a model on one rank only; checkpoint loads (functions.load_model) never communicate."""
    if actifu != "relu":
        raise NotImplementedError("hidden activation other than relu is not used by any reference config")
    if ks != 3 or kernel_ini != "he_normal":
        raise NotImplementedError("ks != 3 / kernel_ini != he_normal are not used by any reference script")
    if dropout_rate_encoder or dropout_rate_decoder or dropout_rate_bottleneck:
        raise NotImplementedError("dropout is 0 in every reference script")
    return UNet(i_height, i_width, i_channels, num_outputmasks, alpha, actifuout, seed=seed, device=device)


def to_numpy_u8(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint8))
