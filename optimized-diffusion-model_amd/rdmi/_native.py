"""ctypes binding of librdmi.so (C ABI: include/rdmi.h).

There is exactly one compute backend: the HIP library built for gfx950.  If it is missing the
import of anything that computes fails loudly -- there is no eager/torch/CPU fallback.
(tests/ may point this module at the CPU *emulator build* of the very same sources with
use_library(); the product never does.)
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT = os.path.join(_HERE, 'librdmi.so')
_lib = None
_lib_path = None

RDMI_PARAMS_CACHED = 1
RDMI_MAX_LEVELS = 8


class Arch(C.Structure):
    _fields_ = [('nf', C.c_int), ('n_levels', C.c_int), ('ch_mult', C.c_int * RDMI_MAX_LEVELS),
                ('num_res_blocks', C.c_int), ('attn_levels', C.c_int), ('channels', C.c_int),
                ('num_classes', C.c_int), ('conditional', C.c_int), ('scale_by_sigma', C.c_int),
                ('fourier_2pi_prescaled', C.c_float), ('compute_dtype', C.c_int)]


class PcOpts(C.Structure):
    _fields_ = [('N', C.c_int), ('eps', C.c_float), ('sigma_min', C.c_double), ('sigma_max', C.c_double),
                ('snr', C.c_float), ('n_steps_each', C.c_int), ('corrector', C.c_int), ('use_cfg', C.c_int),
                ('seed', C.c_uint64), ('seq_offset', C.c_uint64)]


class OdeOpts(C.Structure):
    _fields_ = [('T', C.c_double), ('eps', C.c_double), ('rtol', C.c_double), ('atol', C.c_double), ('sigma_min', C.c_double),
                ('sigma_max', C.c_double), ('moll', C.c_float), ('use_cfg', C.c_int), ('first_step', C.c_double),
                ('max_steps', C.c_int), ('h_next_out', C.POINTER(C.c_double))]


class OptSlot(C.Structure):
    _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p), ('exp_avg', C.c_void_p), ('exp_avg_sq', C.c_void_p),
                ('ema', C.c_void_p), ('numel', C.c_ulonglong)]


class OptHyper(C.Structure):
    _fields_ = [('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float), ('weight_decay', C.c_float),
                ('lr_d', C.c_double), ('beta1_d', C.c_double), ('beta2_d', C.c_double), ('decoupled_wd', C.c_int),
                ('step', C.c_int), ('max_norm', C.c_float), ('ema_decay_d', C.c_double), ('write_back_grad', C.c_int)]


_F = C.c_void_p   # device float* travel as integers (tensor.data_ptr())
_PROTOS = {
    'rdmi_create': ([C.POINTER(Arch), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    'rdmi_destroy': ([C.c_void_p], C.c_int),
    'rdmi_num_params': ([C.c_void_p], C.c_int),
    'rdmi_param_info': ([C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)], C.c_int),
    'rdmi_set_param': ([C.c_void_p, C.c_char_p, _F, C.c_size_t], C.c_int),
    'rdmi_repack': ([C.c_void_p, C.c_void_p], C.c_int),
    'rdmi_forward': ([C.c_void_p, _F, _F, _F, _F, C.c_int, C.c_uint, C.c_void_p], C.c_int),
    'rdmi_score': ([C.c_void_p, _F, _F, _F, _F, C.c_int, C.c_double, C.c_double, C.c_uint, C.c_void_p], C.c_int),
    'rdmi_cf_score': ([C.c_void_p, _F, _F, _F, _F, _F, C.c_int, C.c_double, C.c_double, C.c_uint, C.c_void_p], C.c_int),
    'rdmi_reflect': ([_F, _F, C.c_size_t, C.c_void_p], C.c_int),
    'rdmi_score_hk': ([_F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p], C.c_int),
    'rdmi_perturb': ([_F, _F, _F, _F, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p], C.c_int),
    'rdmi_gto_pack': ([_F, C.c_void_p, _F, _F, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p], C.c_int),
    'rdmi_gto_unnormalize': ([_F, _F, C.c_void_p, C.c_int, C.c_int, C.c_void_p], C.c_int),
    'rdmi_sm_loss': ([_F, _F, _F, _F, _F, _F, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p], C.c_int),
    'rdmi_enable_training': ([C.c_void_p], C.c_int),
    'rdmi_train_forward': ([C.c_void_p, _F, _F, _F, _F, C.c_int, C.c_float, C.c_uint64, C.c_void_p], C.c_int),
    'rdmi_backward': ([C.c_void_p, _F, _F, C.c_size_t, _F, C.c_void_p], C.c_int),
    'rdmi_train_graph_stats': ([C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)], C.c_int),
    'rdmi_em_update': ([_F, _F, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p], C.c_int),
    'rdmi_langevin_update': ([_F, _F, _F, _F, _F, _F, C.c_int, C.c_int, C.c_float, C.c_void_p], C.c_int),
    'rdmi_pc_sample': ([C.c_void_p, _F, _F, _F, _F, _F, _F, C.c_int, C.POINTER(PcOpts), C.c_uint, C.c_void_p], C.c_int),
    'rdmi_ode_sample': ([C.c_void_p, _F, _F, _F, C.c_int, C.POINTER(OdeOpts), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_uint, C.c_void_p], C.c_int),
    'rdmi_opt_create': ([C.POINTER(OptSlot), C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    'rdmi_opt_update_slots': ([C.c_void_p, C.POINTER(OptSlot), C.c_int, C.c_void_p], C.c_int),
    'rdmi_opt_step': ([C.c_void_p, C.POINTER(OptHyper), _F, C.c_void_p], C.c_int),
    'rdmi_opt_destroy': ([C.c_void_p], C.c_int),
    'rdmi_get_tap': ([C.c_void_p, C.c_char_p, _F, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                      C.POINTER(C.c_int), C.c_void_p], C.c_int),
    'rdmi_set_profiling': ([C.c_void_p, C.c_int], C.c_int),
    'rdmi_get_profile': ([C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_long),
                          C.POINTER(C.c_double)], C.c_int),
    'rdmi_path_info': ([C.c_void_p], C.c_char_p),
    'rdmi_debug_op_cycles': ([C.c_void_p, C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_char_p), C.c_int], C.c_int),
    'rdmi_coop_status': ([C.c_void_p, C.POINTER(C.c_int)], C.c_int),
    'rdmi_philox_normal': ([_F, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint, C.c_void_p], C.c_int),
    'rdmi_last_error': ([], C.c_char_p),
    'rdmi_version': ([], C.c_char_p),
}
EXPORTS = tuple(_PROTOS)


def use_library(path):
    """Bind a specific build of the library (tests use this for the emulator build)."""
    global _lib, _lib_path
    lib = C.CDLL(path)
    for name, (argt, rest) in _PROTOS.items():
        fn = getattr(lib, name)       # AttributeError here = a symbol of include/rdmi.h is not exported
        fn.argtypes, fn.restype = argt, rest
    _lib, _lib_path = lib, path
    return lib


def lib():
    if _lib is None:
        if not os.path.exists(_DEFAULT):
            raise RuntimeError(
                f'librdmi.so not found at {_DEFAULT}: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950).  rdmi has no fallback compute path.')
        use_library(_DEFAULT)
    return _lib


def library_path():
    lib()
    return _lib_path


def is_emulator():
    return b'emulator' in lib().rdmi_version()


def check(rc):
    if rc != 0:
        raise RuntimeError('librdmi: ' + lib().rdmi_last_error().decode())


def ptr(t):
    """Device pointer of a dense fp32 tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_contiguous(), 'librdmi takes dense fp32 tensors'
    return t.data_ptr()


def stream_of(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


def philox_normal(n, seed, elem_offset, draw, device):
    """The sampler's own N(0,1) stream (rdmi_philox_normal): `n` values of noise tensor `draw` from global element `elem_offset` on."""
    z = torch.empty(int(n), dtype=torch.float32, device=device)
    require_device(z)
    ctxm = torch.cuda.device(z.device) if z.is_cuda else _Null()
    with ctxm:
        check(lib().rdmi_philox_normal(ptr(z), z.numel(), int(seed), int(elem_offset), int(draw), stream_of(z)))
    return z


def require_device(t):
    """The HIP build only accepts device memory; the emulator build (tests) only host memory."""
    if is_emulator():
        if t.is_cuda:
            raise RuntimeError('emulator build of librdmi takes CPU tensors')
    elif not t.is_cuda:
        raise RuntimeError('librdmi (gfx950) takes tensors on a HIP device; got a CPU tensor and there is no CPU path')


class Context:
    """Owns one rdmi_ctx (workspace + launch plan) for a fixed (arch, H, W, max model batch, device)."""

    def __init__(self, arch, max_batch, H, W, device):
        self.device = torch.device(device)
        self.max_batch, self.H, self.W = max_batch, H, W
        self._h = C.c_void_p()
        self._owner = lib()              # the library that creates the context is the one that destroys it
        with self._guard():
            check(lib().rdmi_create(C.byref(arch), max_batch, H, W, C.byref(self._h)))
        self._bound = {}
        n = lib().rdmi_num_params(self._h)
        self.param_names = []
        for i in range(n):
            nm, ne = C.c_char_p(), C.c_size_t()
            check(lib().rdmi_param_info(self._h, i, C.byref(nm), C.byref(ne)))
            self.param_names.append((nm.value.decode(), ne.value))

    def _guard(self):
        return torch.cuda.device(self.device) if self.device.type == 'cuda' else _Null()

    def bind(self, named_tensors):
        """(Re)bind parameter storage; only pointers that changed are sent."""
        for name, t in named_tensors:
            p = t.data_ptr()
            if self._bound.get(name) != p:
                assert t.dtype == torch.float32 and t.is_contiguous()
                check(lib().rdmi_set_param(self._h, name.encode(), p, t.numel()))
                self._bound[name] = p

    def close(self):
        if self._h:
            self._owner.rdmi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- entry points ---------------------------------------------------------------
    def forward(self, x, sigma, labels, out, flags=0):
        with self._guard():
            check(lib().rdmi_forward(self._h, ptr(x), ptr(sigma), ptr(labels), ptr(out), x.shape[0], flags, stream_of(x)))

    def score(self, x, t, labels, out, smin, smax, flags=0):
        with self._guard():
            check(lib().rdmi_score(self._h, ptr(x), ptr(t), ptr(labels), ptr(out), x.shape[0], smin, smax, flags, stream_of(x)))

    def cf_score(self, x, t, labels, weight, out, smin, smax, flags=0):
        with self._guard():
            check(lib().rdmi_cf_score(self._h, ptr(x), ptr(t), ptr(labels), ptr(weight), ptr(out), x.shape[0], smin, smax,
                                      flags, stream_of(x)))

    def enable_training(self):
        with self._guard():
            check(lib().rdmi_enable_training(self._h))

    def train_forward(self, x, sigma, labels, out, dropout_p, seed):
        with self._guard():
            check(lib().rdmi_train_forward(self._h, ptr(x), ptr(sigma), ptr(labels), ptr(out), x.shape[0], float(dropout_p),
                                           int(seed), stream_of(x)))

    def backward(self, grad_out, grads_flat, x):
        with self._guard():
            check(lib().rdmi_backward(self._h, ptr(grad_out), ptr(grads_flat), grads_flat.numel(), ptr(x), stream_of(x)))

    def train_graph_stats(self):
        """(recordings, replays) of the training step's launch graphs."""
        a, b = C.c_long(), C.c_long()
        check(lib().rdmi_train_graph_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pc_sample(self, x, labels, weight, noise, trace, teacher, opts, flags=0):
        with self._guard():
            check(lib().rdmi_pc_sample(self._h, ptr(x), ptr(labels), ptr(weight), ptr(noise), ptr(trace), ptr(teacher),
                                       x.shape[0], C.byref(opts), flags, stream_of(x)))

    def ode_sample(self, x, labels, weight, opts, flags=0):
        """-> (nfev, t reached, next |h|); x is updated in place."""
        nfev, tfin, hn = C.c_int(), C.c_double(), C.c_double()
        opts.h_next_out = C.pointer(hn)
        with self._guard():
            check(lib().rdmi_ode_sample(self._h, ptr(x), ptr(labels), ptr(weight), x.shape[0], C.byref(opts), C.byref(nfev), C.byref(tfin),
                                        flags, stream_of(x)))
        return nfev.value, tfin.value, hn.value

    def get_tap(self, name, like, nb):
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        buf = torch.empty(self.max_batch * 256 * self.H * self.W * 4, dtype=torch.float32, device=like.device)
        with self._guard():
            check(lib().rdmi_get_tap(self._h, name.encode(), ptr(buf), buf.numel(), C.byref(c), C.byref(h), C.byref(w),
                                     stream_of(like)))
        n = c.value * h.value * w.value
        return buf[:nb * n].reshape(nb, c.value, h.value, w.value).clone()

    def op_cycles(self):
        cyc = (C.c_longlong * 2048)()
        desc = (C.c_char_p * 512)()
        n = lib().rdmi_debug_op_cycles(self._h, cyc, 2048, desc, 512)
        self.fine = [[int(cyc[256 + i * 8 + k]) for k in range(8)] for i in range(min(n, 220))]
        return [(desc[i].decode(), int(cyc[i])) for i in range(n)]

    def path_info(self):
        return lib().rdmi_path_info(self._h).decode()

    def coop_gave_up(self):
        """True if a co-operative launch of this context ever gave up an inter-workgroup wait (synchronises the device)."""
        v = C.c_int()
        with self._guard():
            check(lib().rdmi_coop_status(self._h, C.byref(v)))
        return bool(v.value)

    def set_profiling(self, on):
        check(lib().rdmi_set_profiling(self._h, int(bool(on))))

    def get_profile(self):
        out, i = [], 0
        while True:
            nm, ms, n, fl = C.c_char_p(), C.c_double(), C.c_long(), C.c_double()
            if lib().rdmi_get_profile(self._h, i, C.byref(nm), C.byref(ms), C.byref(n), C.byref(fl)) != 0:
                break
            out.append(dict(kernel=nm.value.decode(), ms=ms.value, launches=n.value, flops=fl.value))
            i += 1
        return out


class OptPlan:
    """One rdmi_opt: the device pointer table of a parameter list (param, grad, exp_avg, exp_avg_sq[, ema shadow])."""

    def __init__(self, params, grads, exp_avgs, exp_avg_sqs, emas, device):
        self.device = torch.device(device)
        self.ptrs = tuple(t.data_ptr() for ts in (params, grads, exp_avgs, exp_avg_sqs, emas or ()) for t in ts)
        n = len(params)
        self.n, self.has_ema = n, bool(emas)
        self.numels = [p.numel() for p in params]
        tab = self._tab = (OptSlot * n)()
        for i in range(n):
            for t in (params[i], grads[i], exp_avgs[i], exp_avg_sqs[i]) + ((emas[i],) if emas else ()):
                assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == params[i].numel()
                require_device(t)
            tab[i].param, tab[i].grad = params[i].data_ptr(), grads[i].data_ptr()
            tab[i].exp_avg, tab[i].exp_avg_sq = exp_avgs[i].data_ptr(), exp_avg_sqs[i].data_ptr()
            tab[i].ema = emas[i].data_ptr() if emas else None
            tab[i].numel = params[i].numel()
        self._h = C.c_void_p()
        self._owner = lib()
        with self._guard():
            check(lib().rdmi_opt_create(tab, n, C.byref(self._h)))
        self._stream_of = params[0]

    def update_grads(self, grads):
        """The gradients of this step live at new addresses: refresh that column of the table with one asynchronous upload."""
        tab = self._tab
        for i, g in enumerate(grads):
            assert g.dtype == torch.float32 and g.is_contiguous() and g.numel() == self.numels[i]
            tab[i].grad = g.data_ptr()
        with self._guard():
            check(self._owner.rdmi_opt_update_slots(self._h, tab, self.n, stream_of(self._stream_of)))

    def _guard(self):
        return torch.cuda.device(self.device) if self.device.type == 'cuda' else _Null()

    def step(self, hyper, total_norm_out=None):
        with self._guard():
            check(self._owner.rdmi_opt_step(self._h, C.byref(hyper), ptr(total_norm_out), stream_of(self._stream_of)))

    def close(self):
        if self._h:
            self._owner.rdmi_opt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ---- stateless elementwise entry points ------------------------------------------------
def reflect(x):
    require_device(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib().rdmi_reflect(ptr(x), ptr(out), x.numel(), stream_of(x)))
    return out


def score_hk(x, x_orig, sigma, efs, refls, min_cutoff):
    require_device(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    B = x.shape[0]
    check(lib().rdmi_score_hk(ptr(x), ptr(x_orig.contiguous()), ptr(sigma.contiguous().float()), ptr(out), B, x.numel() // B,
                              efs, refls, min_cutoff, stream_of(x)))
    return out


def perturb(batch, z, t, smin, smax):
    require_device(batch)
    out = torch.empty_like(batch)
    B = batch.shape[0]
    check(lib().rdmi_perturb(ptr(batch.contiguous()), ptr(z.contiguous()), ptr(t.contiguous().float()), ptr(out), B,
                             batch.numel() // B, smin, smax, stream_of(batch)))
    return out


def gto_pack(data, idx, elems, mean, std):
    """data [rows, L] fp32 device table, idx [B] int64 (or None) -> (images [B, elems], labels [B]); RD/datasets.py:82-98."""
    require_device(data)
    assert data.dim() == 2 and data.dtype == torch.float32 and data.is_contiguous()
    B = data.shape[0] if idx is None else idx.numel()
    if idx is not None:
        assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.device == data.device
    images = torch.empty(B, elems, dtype=torch.float32, device=data.device)
    labels = torch.empty(B, dtype=torch.float32, device=data.device)
    check(lib().rdmi_gto_pack(ptr(data), None if idx is None else idx.data_ptr(), ptr(images), ptr(labels), B, data.shape[1],
                              elems, mean, std, stream_of(data)))
    return images, labels


def gto_unnormalize(samples):
    """samples [N, ...>=67 values per sample] on the device -> (out [N, 67] physical vectors, clip count tensor [1] int64);
    Benchmark/gto_halo_benchmarking.py:255-361."""
    require_device(samples)
    N = samples.shape[0]
    row = 1
    for d in samples.shape[1:]:
        row *= int(d)
    flat = samples.reshape(N, row).contiguous().float()
    out = torch.empty(N, 67, dtype=torch.float32, device=samples.device)
    clips = torch.zeros(1, dtype=torch.int64, device=samples.device)
    check(lib().rdmi_gto_unnormalize(ptr(flat), ptr(out), clips.data_ptr(), N, flat.shape[1], stream_of(samples)))
    return out, clips


def sm_loss(score, perturbed, batch, t, smin, smax, likelihood_weighting, reduce_mean, want_grad=False):
    require_device(batch)
    B = batch.shape[0]
    per = torch.empty(B, dtype=torch.float32, device=batch.device)
    dscore = torch.empty_like(batch, dtype=torch.float32) if want_grad else None
    check(lib().rdmi_sm_loss(ptr(score.contiguous()), ptr(perturbed.contiguous()), ptr(batch.contiguous()),
                             ptr(t.contiguous().float()), ptr(per), ptr(dscore), B, batch.numel() // B, smin, smax,
                             int(bool(likelihood_weighting)), int(bool(reduce_mean)), stream_of(batch)))
    return (per, dscore) if want_grad else per


def em_update(x, score, z, t, N, smin, smax):
    require_device(x)
    x_out, x_mean = torch.empty_like(x), torch.empty_like(x)
    B = x.shape[0]
    check(lib().rdmi_em_update(ptr(x.contiguous()), ptr(score.contiguous()), ptr(z.contiguous()), ptr(t.contiguous()),
                               ptr(x_out), ptr(x_mean), B, x.numel() // B, N, smin, smax, stream_of(x)))
    return x_out, x_mean


def langevin_update(x, score, z, snr):
    require_device(x)
    x_out, x_mean = torch.empty_like(x), torch.empty_like(x)
    B = x.shape[0]
    scratch = torch.empty(2 * B + 2, dtype=torch.float32, device=x.device)
    check(lib().rdmi_langevin_update(ptr(x.contiguous()), ptr(score.contiguous()), ptr(z.contiguous()), ptr(x_out),
                                     ptr(x_mean), ptr(scratch), B, x.numel() // B, snr, stream_of(x)))
    return x_out, x_mean
