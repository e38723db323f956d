"""Host-side mirror of the reference's model objects for the Conformer-CTC path
(asr/models/conformer_blocks.py): `ConformerEncoder`, `StreamingConformerEncoder`, `CTCDecoder`, plus
`ConformerCTC`, the fused encoder + CTCDecoder + greedy pipeline that `test_asr.py::ASR.offline_stt` runs
step by step (test_asr.py:186-200).

Same constructor keywords, `_build()`, `load_weights()`, `__call__(x, training=False)`,
`set_inference_func()` / `.inference`, `summary()` as the Keras models.  All arithmetic happens in
libmi355asr.so (hand-written HIP for gfx950); PyTorch-ROCm is used for device buffers and the stream only.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib
from . import frontend_consts


def _glorot(rng, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def default_weights(names_and_shapes, rng, sample_rate=16000, n_dft=1024, n_mels=80):
    """Keras default initialisers (glorot_uniform kernels, zero biases, gamma=1, beta=0, BN mean 0 / var 1;
    multihead_attention.py:34,37) for every tensor name the handle expects."""
    w = {}
    for name, shape in names_and_shapes:
        leaf = name.rsplit("/", 1)[-1]
        if name == "mel_layer/real_kernels":
            w[name] = frontend_consts.stft_kernels(n_dft)[0]
        elif name == "mel_layer/imag_kernels":
            w[name] = frontend_consts.stft_kernels(n_dft)[1]
        elif name == "mel_layer/freq2mel":
            w[name] = frontend_consts.freq2mel(sample_rate, n_dft, n_mels)
        elif name.startswith("front/mel_layer/"):
            w[name] = {"real_kernels": frontend_consts.stft_kernels(n_dft)[0],
                       "imag_kernels": frontend_consts.stft_kernels(n_dft)[1],
                       "freq2mel": frontend_consts.freq2mel(sample_rate, n_dft, n_mels)}[leaf]
        elif name.startswith("mel_layer/") and name in frontend_consts.leaf_default_weights(n_mels, sample_rate):
            w[name] = frontend_consts.leaf_default_weights(n_mels, sample_rate)[name]
        elif leaf == "embeddings":                  # tf.keras.layers.Embedding default: uniform(-0.05, 0.05)
            w[name] = rng.uniform(-0.05, 0.05, shape).astype(np.float32)
        elif leaf in ("gamma", "moving_variance"):
            w[name] = np.ones(shape, np.float32)
        elif leaf in ("beta", "moving_mean", "bias", "projection_bias"):  # includes the Keras-MHA [H, hs] biases
            w[name] = np.zeros(shape, np.float32)
        elif leaf in ("query_kernel", "key_kernel", "value_kernel", "projection_kernel"):
            # Keras glorot on a 3-D shape: receptive field = prod(shape[:-2])
            rf = int(np.prod(shape[:-2]))
            w[name] = _glorot(rng, shape, shape[-2] * rf, shape[-1] * rf)
        else:
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            w[name] = _glorot(rng, shape, shape[-2] * rf, shape[-1] * rf)
    return w


class _Handle:
    """Owns one `mi355asr_model*` plus its device workspace."""

    def __init__(self, cfg, device):
        self.lib = _lib.lib()
        self.cfg = cfg
        self.device = torch.device(device)
        self.ptr = ctypes.c_void_p()
        create = {_lib.ChunkConfig: self.lib.mi355asr_chunk_create,
                  _lib.TranslatorConfig: self.lib.mi355asr_translator_create}.get(type(cfg), self.lib.mi355asr_create)
        _lib.check(create(ctypes.byref(cfg), ctypes.byref(self.ptr)))
        self._ws = None
        self.built = False

    def __del__(self):
        try:
            if self.ptr:
                self.lib.mi355asr_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------------
    def weight_names(self):
        n = self.lib.mi355asr_num_weights(self.ptr)
        return [self.lib.mi355asr_weight_name(self.ptr, i).decode() for i in range(n)]

    def load(self, weights: dict, strict=True):
        names = set(self.weight_names())
        for name, arr in weights.items():
            if name not in names:
                if strict:
                    raise _lib.Mi355AsrError("unexpected weight %r" % name)
                continue
            a = np.asarray(arr)
            dt = {"float16": 1, "float64": 3}.get(a.dtype.name, 0)        # MI355ASR_DT_*; anything else goes as fp32
            a = np.ascontiguousarray(a if dt else a.astype(np.float32, copy=False))
            dims = (ctypes.c_int64 * a.ndim)(*a.shape)
            _lib.check(self.lib.mi355asr_load_weight_typed(self.ptr, name.encode(), a.ctypes.data_as(ctypes.c_void_p),
                                                           dt, a.ndim, dims))

    def set_expected_rows(self, rows):
        """Before finalize(): the most rows (batch x encoder frames) a call will bring; below the ring kernels' crossover
        (1 500) the dmodel-256 / 512 dense layers are not packed a second time as slab rings.  None / negative = unknown."""
        _lib.check(self.lib.mi355asr_set_expected_rows(self.ptr, -1 if rows is None else int(rows)))

    def finalize(self):
        if self.device.type != "cuda":
            raise _lib.Mi355AsrError("mi355asr needs a ROCm device (got %s); there is no CPU path" % self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mi355asr_finalize_weights(self.ptr, self._stream()))
        self.built = True

    # ---- device plumbing -----------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def ws_for_wave(self, B, L):
        n = ctypes.c_size_t()
        _lib.check(self.lib.mi355asr_workspace_bytes(self.ptr, B, L, ctypes.byref(n)))
        return self.workspace(n.value), n.value

    def ws_for_frames(self, B, T):
        n = ctypes.c_size_t()
        _lib.check(self.lib.mi355asr_ctc_workspace_bytes(self.ptr, B, T, ctypes.byref(n)))
        return self.workspace(n.value), n.value

    def out_frames(self, L):
        f, t = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self.lib.mi355asr_out_frames(self.ptr, L, ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def to_device(self, x, dtype=torch.float32):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        return x.to(device=self.device, dtype=dtype).contiguous()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()


def _wave2d(h, inputs):
    x = h.to_device(inputs)
    if x.dim() == 3:
        if x.shape[-1] != 1:
            raise ValueError("expected mono input [B, L, 1] (time_frequency.py:71-73), got %s" % (tuple(x.shape),))
        x = x.reshape(x.shape[0], x.shape[1])
    if x.dim() != 2:
        raise ValueError("expected waveform of shape [B, L, 1] or [B, L], got %s" % (tuple(x.shape),))
    return x.contiguous()


def _gemm_dtype(v):
    try:
        return {"float32": 0, "fp32": 0, 0: 0, "bfloat16": 1, "bf16": 1, 1: 1}[v]
    except KeyError:
        raise ValueError("gemm_dtype must be 'float32' or 'bfloat16', got %r" % (v,))


class _ModelBase:
    def _names_and_shapes(self):
        shapes = self._expected_shapes()
        return [(n, shapes[n]) for n in self._h.weight_names()]

    def _build(self, seed=0):
        """Keras `_build()` materialises default-initialised weights; so does this (seeded)."""
        rng = np.random.default_rng(seed)
        w = default_weights(self._names_and_shapes(), rng, self.sample_rate, 1024, self.n_mels)
        self._weights = w
        self._h.load(w)
        self._h.finalize()
        return self

    # tensors that a checkpoint may legitimately leave out: the fixed DFT kernels / mel filterbank of the frontend
    # (non-trainable variables; Keras-default values are generated by _build())
    _DEFAULTABLE = ("mel_layer/real_kernels", "mel_layer/imag_kernels", "mel_layer/freq2mel")

    def load_weights(self, weights, by_name=True, strict=None, allow_missing=False):
        """`weights`: a dict name -> array, the path of an .npz of Keras-layout tensors, the path of a Keras `.h5`
        weight file as the reference's trainers write them (`ctc_runners.py:272-325`; read by the pure-Python HDF5 reader
        h5lite.py, variable names mapped by checkpoint.keras_names_to_abi), a TensorFlow checkpoint prefix / directory, or
        the path of the reference's tf2onnx export of the CTCDecoder (`ctc_model.onnx`; checkpoint.py).

        Loading from a PATH checks coverage: if no tensor of the file maps onto this model, or if tensors of the model
        are absent from it (other than the frontend's fixed DFT / mel matrices), the call raises instead of leaving
        those tensors at their `_build()` values -- a checkpoint whose names fail to map must not "load" as a
        random-weight model.  `allow_missing=True` restores Keras' by_name leniency."""
        from_path = isinstance(weights, (str, os.PathLike))
        if from_path:
            path = str(weights)
            if path.endswith((".h5", ".hdf5")):
                from . import checkpoint
                weights = checkpoint.keras_h5_to_abi(path)
            elif path.endswith(".index") or os.path.isdir(path) or os.path.exists(path + ".index"):
                from . import checkpoint                             # TensorFlow tensor-bundle checkpoint
                weights = (checkpoint.chunk_checkpoint_to_abi(path) if type(self).__name__ == "ChunkConformer"
                           else checkpoint.tf_checkpoint_to_abi(path))
            elif path.endswith(".onnx"):
                from . import checkpoint
                weights = checkpoint.ctc_decoder_weights_from_onnx(path, num_heads=self.num_heads)
            else:
                weights = dict(np.load(path))
            expected = set(self._h.weight_names())
            provided = expected & set(weights)
            if not provided:
                raise _lib.Mi355AsrError("%s: none of the %d tensors in this file maps onto %s (first names: %s)"
                                         % (path, len(weights), self.name, sorted(weights)[:3]))
            missing = sorted(n for n in expected - provided if not n.endswith(self._DEFAULTABLE))
            if missing and not allow_missing:
                raise _lib.Mi355AsrError("%s: %d tensors of %s are not in this file (e.g. %s); pass allow_missing=True to keep "
                                         "their initial values" % (path, len(missing), self.name, missing[:4]))
        if strict is None:
            strict = not by_name
        if not getattr(self, "_weights", None):
            self._build()
        self._weights.update({k: np.asarray(v, np.float32) for k, v in weights.items() if k in self._weights or strict})
        self._h.load(weights, strict=strict)
        self._h.finalize()
        return self

    def get_weights_dict(self):
        return dict(self._weights)

    def count_params(self):
        return int(sum(int(np.prod(s)) for _, s in self._names_and_shapes()))

    def summary(self, line_length=100):
        print("=" * line_length)
        print("%s  (libmi355asr: %s)" % (self.name, self._h.lib.mi355asr_version().decode()))
        for n, s in self._names_and_shapes():
            print("%-*s %s" % (line_length - 20, n, tuple(s)))
        print("Total params: {:,}".format(self.count_params()))
        print("=" * line_length)

    def set_inference_func(self):
        self.inference = lambda *a, **k: self.__call__(*a, training=False, **k)


def _block_shapes(p, d, H, hs, k):
    s = {}
    for ff in ("ff_module_1", "ff_module_2"):
        q = "%s/%s" % (p, ff)
        s[q + "/ln/gamma"] = (d,)
        s[q + "/ln/beta"] = (d,)
        s[q + "/ffn1/kernel"] = (d, 4 * d)
        s[q + "/ffn1/bias"] = (4 * d,)
        s[q + "/ffn2/kernel"] = (4 * d, d)
        s[q + "/ffn2/bias"] = (d,)
    m = p + "/mhsa_module"
    s[m + "/ln/gamma"] = (d,)
    s[m + "/ln/beta"] = (d,)
    for nm in ("query_kernel", "key_kernel", "value_kernel"):
        s[m + "/mha/" + nm] = (H, d, hs)
    s[m + "/mha/projection_kernel"] = (H, hs, d)
    s[m + "/mha/projection_bias"] = (d,)
    c = p + "/conv_module"
    s[c + "/ln/gamma"] = (d,)
    s[c + "/ln/beta"] = (d,)
    s[c + "/pw_conv_1/kernel"] = (1, d, 2 * d)
    s[c + "/pw_conv_1/bias"] = (2 * d,)
    s[c + "/dw_conv/depthwise_kernel"] = (k, d, 1)
    s[c + "/dw_conv/pointwise_kernel"] = (1, d, 2 * d)
    s[c + "/dw_conv/bias"] = (2 * d,)
    for nm in ("gamma", "beta", "moving_mean", "moving_variance"):
        s[c + "/bn/" + nm] = (2 * d,)
    s[c + "/pw_conv_2/kernel"] = (1, 2 * d, d)
    s[c + "/pw_conv_2/bias"] = (d,)
    s[p + "/ln/gamma"] = (d,)
    s[p + "/ln/beta"] = (d,)
    return s


def _wav_layer_shapes(d):
    """WavePickModel variables (asr/models/wav_model.py:108-131) under the handle's `wav_layer/` names."""
    s = {"wav_layer/sep_conv/depthwise_kernel": (7, 1, 1), "wav_layer/sep_conv/pointwise_kernel": (1, 1, 32),
         "wav_layer/sep_conv/bias": (32,)}
    cin = 32
    for i in range(1, 4):
        c = min(32 * (i + 1), d)
        s["wav_layer/conv_%d/kernel" % i], s["wav_layer/conv_%d/bias" % i] = (3, cin, c), (c,)
        for sub, k in (("conv5", 5), ("conv1", 1), ("shortcut", 1)):
            s["wav_layer/res_%d/%s/kernel" % (i, sub)], s["wav_layer/res_%d/%s/bias" % (i, sub)] = (k, c, c), (c,)
        cin = c
    s["wav_layer/final/kernel"], s["wav_layer/final/bias"] = (7, cin, d), (d,)
    return s


_MEL_LAYER_CODE = {"Melspectrogram": 0, "leaf": 1, "Spectrogram": 2}      # mi355asr_config.mel_layer_type


def _encoder_shapes(d, H, hs, k, num_blocks, n_mels, n_dft=1024, leaf=False, add_wav_info=False, spectrogram=False):
    nb = n_dft // 2 + 1
    if spectrogram:                      # the plain Spectrogram layer: all nb dB bins go to the subsampling convs
        n_mels = nb
    f2 = -(-(-(-n_mels // 2)) // 2)
    if leaf:
        s = {"mel_layer/tfbanks_preemp/kernel": (2, 1, 1), "mel_layer/tfbanks_complex_conv/kernel": (n_mels, 2),
             "mel_layer/learnable_pooling/kernel": (1, 1, n_mels, 1), "mel_layer/PCEN/alpha": (n_mels,),
             "mel_layer/PCEN/delta": (n_mels,), "mel_layer/PCEN/root": (n_mels,), "mel_layer/PCEN/EMA/smooth": (n_mels,),
             "mel_layer/tfbanks_instancenorm/gamma": (n_mels,), "mel_layer/tfbanks_instancenorm/beta": (n_mels,)}
    else:
        s = {"mel_layer/real_kernels": (n_dft, 1, 1, nb), "mel_layer/imag_kernels": (n_dft, 1, 1, nb)}
        if not spectrogram:
            s["mel_layer/freq2mel"] = (nb, n_mels)
    s.update({
         "conv_subsampling/conv1/kernel": (3, 3, 1, d), "conv_subsampling/conv1/bias": (d,),
         "conv_subsampling/conv2/kernel": (3, 3, d, d), "conv_subsampling/conv2/bias": (d,),
         "conv_subsampling/linear/kernel": (f2 * d, d), "conv_subsampling/linear/bias": (d,)})
    if add_wav_info:
        s.update(_wav_layer_shapes(d))
    for i in range(num_blocks):
        s.update(_block_shapes("conformer_block_%d" % i, d, H, hs, k))
    return s


def _ctc_shapes(d, H, hs, k, num_blocks, num_classes):
    s = {"project/kernel": (d, d), "project/bias": (d,)}
    for i in range(num_blocks):
        s.update(_block_shapes("decoder_conformer_block_%d" % i, d, H, hs, k))
    s["fully_connected/kernel"] = (d, num_classes)
    s["fully_connected/bias"] = (num_classes,)
    return s


class ConformerEncoder(_ModelBase):
    """asr/models/conformer_blocks.py:277-384.  `mel_layer_type`: 'Melspectrogram' (the default of
    asr/configs/am_data.yml:3), 'leaf', or -- as in the reference, any other value -- the plain 'Spectrogram' layer
    (conformer_blocks.py:318-323: 513 dB bins, no mel matrix); `add_wav_info=True` adds the WavePickModel branch
    (wav_model.py:108-146) to the subsampled features."""

    def __init__(self, dmodel=144, reduction_factor=4, num_blocks=16, head_size=36, num_heads=4, kernel_size=32,
                 fc_factor=0.5, dropout=0.0, add_wav_info=False, sample_rate=16000, n_mels=80,
                 mel_layer_type="leaf", mel_layer_trainable=False, stride_ms=10, name="conformer_encoder",
                 device="cuda:0", chunk_size=0, gemm_dtype="float32", **kwargs):
        """gemm_dtype (not in the reference): "float32" (default, the reference's arithmetic) or "bfloat16" = bf16 MFMA
        inputs with fp32 accumulation for the dense layers (BASELINE config 3)."""
        self.gemm_dtype = _gemm_dtype(gemm_dtype)
        self.mel_layer_type = mel_layer_type if mel_layer_type in ("Melspectrogram", "leaf") else "Spectrogram"
        self.name = name
        self.dmodel, self.num_heads, self.head_size = dmodel, num_heads, head_size
        self.fc_factor, self.dropout = fc_factor, dropout      # dropout is identity at inference
        self.reduction_factor = reduction_factor
        self.num_blocks, self.kernel_size = num_blocks, kernel_size
        self.sample_rate, self.n_mels, self.stride_ms = sample_rate, n_mels, stride_ms
        self.hop_size = int(stride_ms * sample_rate // 1000) * reduction_factor   # conformer_blocks.py:302
        self.add_wav_info = bool(add_wav_info)
        self.chunk_size = int(chunk_size)
        self._device = device
        self._weights = None
        self._make_handle()

    def _make_handle(self):
        cfg = _lib.Config(dmodel=self.dmodel, num_blocks=self.num_blocks, head_size=self.head_size,
                          num_heads=self.num_heads, kernel_size=self.kernel_size, fc_factor=self.fc_factor,
                          reduction_factor=self.reduction_factor, n_mels=self.n_mels, sample_rate=self.sample_rate,
                          stride_ms=self.stride_ms, n_dft=1024, chunk_size=self.chunk_size, has_encoder=1,
                          num_classes=0, ctc_num_blocks=0, ctc_kernel_size=32, ctc_fc_factor=0.5,
                          gemm_dtype=self.gemm_dtype, mel_layer_type=_MEL_LAYER_CODE[self.mel_layer_type],
                          add_wav_info=int(self.add_wav_info))
        self._h = _Handle(cfg, self._device)

    def _expected_shapes(self):
        return _encoder_shapes(self.dmodel, self.num_heads, self.head_size, self.kernel_size, self.num_blocks,
                               self.n_mels, leaf=self.mel_layer_type == "leaf", add_wav_info=self.add_wav_info,
                               spectrogram=self.mel_layer_type == "Spectrogram")

    def __call__(self, inputs, training=False, **kwargs):
        """wav [B, L, 1] (or [B, L]) float32 -> torch.Tensor [B, T, dmodel] on the device."""
        if training:
            raise NotImplementedError("inference path only (training=False)")
        h = self._h
        if not h.built:
            self._build()
        x = _wave2d(h, inputs)
        B, L = x.shape
        _, T = h.out_frames(L)
        out = torch.empty((B, T, self.dmodel), dtype=torch.float32, device=h.device)
        ws, n = h.ws_for_wave(B, L)
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_encoder_forward(h.ptr, _p(x), B, L, _p(out), _p(ws), n, h._stream()))
        return out

    # stage-level access (tests localise mismatches with these)
    def melspectrogram(self, inputs):
        h = self._h
        x = _wave2d(h, inputs)
        B, L = x.shape
        F, _ = h.out_frames(L)
        nblk = L // self.chunk_size if self.chunk_size else 1
        nm = 513 if self.mel_layer_type == "Spectrogram" else self.n_mels      # n_dft / 2 + 1 bins without the mel matrix
        out = torch.empty((B * nblk, F // nblk, nm), dtype=torch.float32, device=h.device)
        ws, n = h.ws_for_wave(B, L)
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_melspectrogram(h.ptr, _p(x), B, L, _p(out), _p(ws), n, h._stream()))
        return out

    def conv_subsampling(self, mel):
        h = self._h
        m = h.to_device(mel)
        B, F, _ = m.shape
        T = -(-(-(-F // (self.reduction_factor // 2))) // 2)
        out = torch.empty((B, T, self.dmodel), dtype=torch.float32, device=h.device)
        hop = int(self.stride_ms * self.sample_rate // 1000)
        ws, n = h.ws_for_wave(B, self.chunk_size if self.chunk_size else F * hop)
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_conv_subsampling(h.ptr, _p(m), B, F, _p(out), _p(ws), n, h._stream()))
        return out

    def conformer_block(self, index, x, stack=0):
        h = self._h
        xd = h.to_device(x)
        B, T, _ = xd.shape
        out = torch.empty_like(xd)
        ws, n = h.ws_for_frames(B, T)
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_conformer_block(h.ptr, stack, index, _p(xd), B, T, _p(out), _p(ws), n, h._stream()))
        return out


class StreamingConformerEncoder(ConformerEncoder):
    """conformer_blocks.py:567-614 (Block Conformer): the waveform is cut into `chunk_size`-sample blocks that
    go through the ordinary encoder as independent batch entries; outputs are concatenated in time."""

    def __init__(self, *args, name="stream_conformer_encoder", **kwargs):
        super().__init__(*args, name=name, **kwargs)

    def add_chunk_size(self, chunk_size, mel_size, hop_size):
        self.chunk_size = int(chunk_size)
        self.mel_size = mel_size
        self.mel_length = chunk_size // hop_size if chunk_size % hop_size == 0 else chunk_size // hop_size + 1
        w = self._weights
        self._make_handle()
        if w is not None:
            self._h.load(w)
            self._h.finalize()

    def set_inference_func(self):
        # The reference's override never assigns self.inference (conformer_blocks.py:596-614); the intended
        # behaviour is the base-class inference applied to one block at a time.
        self.inference = lambda x, **k: self.__call__(x, training=False)


class CTCDecoder(_ModelBase):
    """asr/models/conformer_blocks.py:385-438: Dense(d->d) + num_blocks ConformerBlocks + Dense(d->num_classes)."""

    def __init__(self, num_classes, dmodel=144, num_blocks=16, head_size=36, num_heads=4, fc_factor=0.5,
                 dropout=0.0, kernel_size=32, device="cuda:0", name="ctc_decoder", gemm_dtype="float32", **kwargs):
        self.name = name
        self.num_classes, self.dmodel = num_classes, dmodel
        self.num_blocks, self.head_size, self.num_heads = num_blocks, head_size, num_heads
        self.fc_factor, self.kernel_size = fc_factor, kernel_size
        self.sample_rate, self.n_mels = 16000, 80
        self._weights = None
        cfg = _lib.Config(dmodel=dmodel, num_blocks=0, head_size=head_size, num_heads=num_heads, kernel_size=32,
                          fc_factor=0.5, reduction_factor=4, n_mels=80, sample_rate=16000, stride_ms=10, n_dft=1024,
                          chunk_size=0, has_encoder=0, num_classes=num_classes, ctc_num_blocks=num_blocks,
                          ctc_kernel_size=kernel_size, ctc_fc_factor=fc_factor, gemm_dtype=_gemm_dtype(gemm_dtype))
        self._h = _Handle(cfg, device)

    def _expected_shapes(self):
        return _ctc_shapes(self.dmodel, self.num_heads, self.head_size, self.kernel_size, self.num_blocks,
                           self.num_classes)

    def __call__(self, inputs, training=None, mask=None, return_argmax=False, return_logits=True):
        """enc [B, T, dmodel] -> logits [B, T, num_classes] (torch, on device).  return_argmax: (logits, per-frame argmax);
        with return_logits=False the logits are never written (mi355asr_ctc_forward takes NULL: the class head keeps only its
        running argmax -- what a greedy decode needs) and the first element is None."""
        if training:
            raise NotImplementedError("inference path only")
        h = self._h
        if not h.built:
            self._build()
        x = h.to_device(inputs)
        B, T, _ = x.shape
        if not return_logits and not return_argmax:
            raise ValueError("nothing to return: return_logits=False needs return_argmax=True")
        logits = torch.empty((B, T, self.num_classes), dtype=torch.float32, device=h.device) if return_logits else None
        amax = torch.empty((B, T), dtype=torch.int32, device=h.device)
        ws, n = h.ws_for_frames(B, T)
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_ctc_forward(h.ptr, _p(x), B, T, _p(logits) if return_logits else None, _p(amax), _p(ws), n, h._stream()))
        return (logits, amax) if return_argmax else logits

    def conformer_block(self, index, x):
        return ConformerEncoder.conformer_block(self, index, x, stack=1)


class Translator(_ModelBase):
    """asr/models/conformer_blocks.py:505-566: Embedding(inp_classes -> d) + num_blocks RBlocks (FFModule, cross-
    attention of the token stream over the encoder output with a sinusoidal positional term on the query,
    ConvModule, FFModule, LayerNorm) + Dense(d -> tar_classes).  Called as the reference calls it:
    `translator([ctc_decode, enc_outputs], training=False)` (test_asr.py:202) or `.inference(ids, enc)` (:149)."""

    def __init__(self, inp_classes, tar_classes, dmodel=144, num_blocks=16, head_size=36, num_heads=4,
                 fc_factor=0.5, dropout=0.0, kernel_size=32, device="cuda:0", name="translator", **kwargs):
        self.name = name
        self.inp_classes, self.tar_classes, self.dmodel = inp_classes, tar_classes, dmodel
        self.num_blocks, self.head_size, self.num_heads = num_blocks, head_size, num_heads
        self.fc_factor, self.kernel_size = fc_factor, kernel_size
        self.sample_rate, self.n_mels = 16000, 80
        self._weights = None
        cfg = _lib.TranslatorConfig(dmodel=dmodel, num_blocks=num_blocks, head_size=head_size, num_heads=num_heads,
                                    kernel_size=kernel_size, fc_factor=fc_factor, inp_classes=inp_classes,
                                    tar_classes=tar_classes)
        self._h = _Handle(cfg, device)

    def _expected_shapes(self):
        d = self.dmodel
        s = {"inp_embedding/embeddings": (self.inp_classes, d)}
        for i in range(self.num_blocks):
            s.update(_block_shapes("decoder_conformer_block_%d" % i, d, self.num_heads, self.head_size, self.kernel_size))
        s["fully_connected/kernel"] = (d, self.tar_classes)
        s["fully_connected/bias"] = (self.tar_classes,)
        return s

    def __call__(self, x, training=None, mask=None, return_argmax=False, return_logits=True):
        """x = [ids int [B, U], enc float [B, T, dmodel]] -> logits [B, U, tar_classes] (torch, on device).  return_argmax:
        (logits, per-token argmax); with return_logits=False the logits are never written (the class head keeps its running
        argmax only -- what offline_stt consumes, test_asr.py:203-205): (None, argmax)."""
        if training:
            raise NotImplementedError("inference path only")
        ids, enc = x
        return self._forward(ids, enc, return_argmax, return_logits)

    def set_inference_func(self):
        self.inference = lambda inputs, enc: self._forward(inputs, enc, False)

    def _forward(self, ids, enc, return_argmax, return_logits=True):
        h = self._h
        if not h.built:
            self._build()
        idt = h.to_device(ids, dtype=torch.int32)
        e = h.to_device(enc)
        if idt.dim() != 2 or e.dim() != 3 or e.shape[0] != idt.shape[0] or e.shape[2] != self.dmodel:
            raise ValueError("expected ids [B, U] and enc [B, T, %d], got %s and %s"
                             % (self.dmodel, tuple(idt.shape), tuple(e.shape)))
        B, U = idt.shape
        T = e.shape[1]
        if U == 0:                                   # nothing decoded: Keras returns an empty [B, 0, V] tensor
            z = torch.empty((B, 0, self.tar_classes), dtype=torch.float32, device=h.device)
            return (z, torch.empty((B, 0), dtype=torch.int32, device=h.device)) if return_argmax else z
        if not return_logits and not return_argmax:
            raise ValueError("nothing to return: return_logits=False needs return_argmax=True")
        logits = torch.empty((B, U, self.tar_classes), dtype=torch.float32, device=h.device) if return_logits else None
        amax = torch.empty((B, U), dtype=torch.int32, device=h.device)
        n = ctypes.c_size_t()
        _lib.check(h.lib.mi355asr_translator_workspace_bytes(h.ptr, B, U, T, ctypes.byref(n)))
        ws = h.workspace(n.value)
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_translator_forward(h.ptr, _p(idt), _p(e), B, U, T, _p(logits) if return_logits else None, _p(amax),
                                                         _p(ws), n.value, h._stream()))
        return (logits, amax) if return_argmax else logits


def ctc_greedy_decode(frame_argmax, input_length=None, blank=None, device=None):
    """tf.keras.backend.ctc_decode(greedy=True)[0][0] on per-frame argmax ids: merge repeated, drop blank,
    pad with -1 (test_asr.py:196-200).  frame_argmax int32 [B,T] -> (ids int32 [B,T], lengths int32 [B])."""
    lib = _lib.lib()
    fa = frame_argmax if torch.is_tensor(frame_argmax) else torch.as_tensor(np.asarray(frame_argmax))
    dev = torch.device(device) if device is not None else (fa.device if fa.is_cuda else torch.device("cuda:0"))
    fa = fa.to(device=dev, dtype=torch.int32).contiguous()
    B, T = fa.shape
    il = None
    if input_length is not None:
        il = torch.as_tensor(np.asarray(input_length) if not torch.is_tensor(input_length) else input_length)
        il = il.to(device=dev, dtype=torch.int32).contiguous()
    ids = torch.empty((B, T), dtype=torch.int32, device=dev)
    lens = torch.empty((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.mi355asr_ctc_greedy(_p(fa), _p(il), B, T, int(blank), _p(ids), _p(lens), st))
    return ids, lens


def frame_argmax(x, device=None):
    """per-frame argmax of logits / probabilities [..., V] on the device (first maximum wins) -> int32 [...]"""
    lib = _lib.lib()
    t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
    dev = torch.device(device) if device is not None else (t.device if t.is_cuda else torch.device("cuda:0"))
    t = t.to(device=dev, dtype=torch.float32).contiguous()
    V = t.shape[-1]
    out = torch.empty(t.shape[:-1], dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.mi355asr_frame_argmax(_p(t), int(out.numel()), int(V), _p(out), st))
    return out


class ConformerCTC(_ModelBase):
    """Encoder + CTCDecoder + greedy decode in one handle / one call (`mi355asr_recognize`): the timed region
    of the benchmark and what `ASR.offline_stt` (test_asr.py:186-200) computes up to the token ids."""

    def __init__(self, num_classes, dmodel=144, reduction_factor=4, num_blocks=13, head_size=36, num_heads=4,
                 kernel_size=32, fc_factor=0.5, sample_rate=16000, n_mels=80, stride_ms=10, chunk_size=0,
                 ctcdecoder_num_blocks=1, ctcdecoder_kernel_size=32, ctcdecoder_fc_factor=0.5,
                 device="cuda:0", name="conformer_ctc", gemm_dtype="float32", mel_layer_type="Melspectrogram",
                 add_wav_info=False, **kwargs):
        self.name = name
        self.add_wav_info = bool(add_wav_info)
        # as in the reference (conformer_blocks.py:309-323): any other value selects the plain Spectrogram layer
        self.mel_layer_type = mel_layer_type = mel_layer_type if mel_layer_type in ("Melspectrogram", "leaf") else "Spectrogram"
        self.num_classes, self.dmodel = num_classes, dmodel
        self.blank = num_classes - 1               # utils/text_featurizers.py:65-70 (blank_at_zero: False)
        self.num_blocks, self.head_size, self.num_heads, self.kernel_size = num_blocks, head_size, num_heads, kernel_size
        self.ctc_blocks, self.ctc_kernel = ctcdecoder_num_blocks, ctcdecoder_kernel_size
        self.sample_rate, self.n_mels, self.stride_ms = sample_rate, n_mels, stride_ms
        self.reduction_factor, self.chunk_size = reduction_factor, int(chunk_size)
        self._weights = None
        cfg = _lib.Config(dmodel=dmodel, num_blocks=num_blocks, head_size=head_size, num_heads=num_heads,
                          kernel_size=kernel_size, fc_factor=fc_factor, reduction_factor=reduction_factor,
                          n_mels=n_mels, sample_rate=sample_rate, stride_ms=stride_ms, n_dft=1024,
                          chunk_size=self.chunk_size, has_encoder=1, num_classes=num_classes,
                          ctc_num_blocks=ctcdecoder_num_blocks, ctc_kernel_size=ctcdecoder_kernel_size,
                          ctc_fc_factor=ctcdecoder_fc_factor, gemm_dtype=_gemm_dtype(gemm_dtype),
                          mel_layer_type=_MEL_LAYER_CODE[mel_layer_type], add_wav_info=int(self.add_wav_info))
        self._h = _Handle(cfg, device)

    @classmethod
    def from_config(cls, config, num_classes, device="cuda:0"):
        """config: the merged am_data.yml + model yml dict (utils/user_config.py)."""
        mc, sc = config["model_config"], config["speech_config"]
        chunk = int(sc["streaming_bucket"] * sc["sample_rate"]) if sc.get("streaming") else 0
        return cls(num_classes, dmodel=mc["dmodel"], reduction_factor=mc["reduction_factor"],
                   num_blocks=mc["num_blocks"], head_size=mc["head_size"], num_heads=mc["num_heads"],
                   kernel_size=mc["kernel_size"], fc_factor=mc["fc_factor"], sample_rate=sc["sample_rate"],
                   n_mels=sc["num_feature_bins"], stride_ms=sc["stride_ms"], chunk_size=chunk,
                   ctcdecoder_num_blocks=mc["ctcdecoder_num_blocks"],
                   ctcdecoder_kernel_size=mc["ctcdecoder_kernel_size"],
                   ctcdecoder_fc_factor=mc["ctcdecoder_fc_factor"], device=device)

    def _expected_shapes(self):
        s = _encoder_shapes(self.dmodel, self.num_heads, self.head_size, self.kernel_size, self.num_blocks, self.n_mels,
                            leaf=self.mel_layer_type == "leaf", add_wav_info=self.add_wav_info,
                            spectrogram=self.mel_layer_type == "Spectrogram")
        s.update(_ctc_shapes(self.dmodel, self.num_heads, self.head_size, self.ctc_kernel, self.ctc_blocks,
                             self.num_classes))
        return s

    def out_frames(self, L):
        return self._h.out_frames(L)[1]

    def prepare(self, B, L):
        """pre-allocate workspace + outputs for a fixed [B, L] so that recognize() does no allocation."""
        h = self._h
        T = h.out_frames(L)[1]
        ws, n = h.ws_for_wave(B, L)
        self._ids = torch.empty((B, T), dtype=torch.int32, device=h.device)
        self._lens = torch.empty((B,), dtype=torch.int32, device=h.device)
        return T

    def recognize(self, wav, input_length=None, reuse_buffers=False, out=None):
        """wav [B,L(,1)] on device -> (ids int32 [B,T] padded -1, lengths int32 [B]).  Asynchronous.
        reuse_buffers=True returns the model's own pre-allocated output tensors (no allocation, no copy -- what the
        C-ABI call writes into); the NEXT recognize() of the same shape overwrites them, so only use it when the
        results are consumed before the next call.  out=(ids, lens): the C-ABI call writes into these contiguous int32
        device tensors of shapes [B, T] and [B] instead (a caller that keeps several batches in flight, e.g. while
        their ids travel over RCCL, rotates its own buffers)."""
        h = self._h
        if not h.built:
            self._build()
        x = _wave2d(h, wav)
        B, L = x.shape
        T = h.out_frames(L)[1]
        ws, n = h.ws_for_wave(B, L)
        ids = getattr(self, "_ids", None)
        if ids is None or tuple(ids.shape) != (B, T):
            self.prepare(B, L)
        il = None
        if input_length is not None:
            il = h.to_device(input_length, torch.int32)
        o_ids, o_lens = (self._ids, self._lens) if out is None else out
        if out is not None and not (o_ids.is_cuda and o_ids.dtype == torch.int32 and o_ids.is_contiguous() and
                                    tuple(o_ids.shape) == (B, T) and o_lens.is_cuda and o_lens.dtype == torch.int32 and
                                    o_lens.is_contiguous() and tuple(o_lens.shape) == (B,)):
            raise ValueError("out=(ids, lens): contiguous int32 device tensors of shapes [%d, %d] and [%d]" % (B, T, B))
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_recognize(h.ptr, _p(x), B, L, _p(il), _p(o_ids), _p(o_lens), _p(ws), n,
                                                h._stream()))
        if reuse_buffers or out is not None:
            return o_ids, o_lens
        return self._ids.clone(), self._lens.clone()

    __call__ = recognize

    def encode(self, wav):
        h = self._h
        x = _wave2d(h, wav)
        B, L = x.shape
        T = h.out_frames(L)[1]
        out = torch.empty((B, T, self.dmodel), dtype=torch.float32, device=h.device)
        ws, n = h.ws_for_wave(B, L)
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_encoder_forward(h.ptr, _p(x), B, L, _p(out), _p(ws), n, h._stream()))
        return out

    def ctc_logits(self, enc, return_argmax=False):
        h = self._h
        x = h.to_device(enc)
        B, T, _ = x.shape
        logits = torch.empty((B, T, self.num_classes), dtype=torch.float32, device=h.device)
        amax = torch.empty((B, T), dtype=torch.int32, device=h.device)
        ws, n = h.ws_for_frames(B, T)
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_ctc_forward(h.ptr, _p(x), B, T, _p(logits), _p(amax), _p(ws), n, h._stream()))
        return (logits, amax) if return_argmax else logits

    melspectrogram = ConformerEncoder.melspectrogram
    conv_subsampling = ConformerEncoder.conv_subsampling

    def conformer_block(self, index, x, stack=0):
        return ConformerEncoder.conformer_block(self, index, x, stack=stack)


def ctc_prefix_beam_decode(x, input_length=None, beam_width=10, cutoff_prob=0.99, cutoff_top_n=40, is_logits=False,
                           num_threads=None, max_len=None):
    """Scorer-less CTC prefix beam search of externals/ctc_decoders (ctc_beam_search_decoder_batch).

    x: [B, T, V] probabilities (or logits with is_logits=True), blank = class V-1.  A CUDA tensor goes through the
    GPU top-n selection kernel + host search (`mi355asr_ctc_prefix_beam`, needs cutoff_prob < 1); a NumPy array / CPU
    tensor of probabilities runs entirely on the host threads (`mi355asr_ctc_prefix_beam_host`).
    -> (ids int32 [B, beam, max_len] padded -1, lens int32 [B, beam], scores float32 [B, beam], n_hyp int32 [B])."""
    lib = _lib.lib()
    on_gpu = torch.is_tensor(x) and x.is_cuda
    B, T, V = x.shape
    if T == 0 and B > 0:
        # no frames at all (ChunkConformer: a batch in which the phone picker kept nothing): every utterance decodes to the
        # empty prefix with log-probability 0, as T_u = 0 does inside a batch (the C entry points want T >= 1)
        ml = int(max_len or 1)
        lens = np.zeros((B, beam_width), np.int32)
        scores = np.full((B, beam_width), -np.finfo(np.float32).max, np.float32)
        scores[:, 0] = 0.0
        return np.full((B, beam_width, ml), -1, np.int32), lens, scores, np.ones((B,), np.int32)
    max_len = int(max_len or T)
    nthreads = int(num_threads or min(B, os.cpu_count() or 1))
    ids = np.empty((B, beam_width, max_len), np.int32)
    lens = np.empty((B, beam_width), np.int32)
    scores = np.empty((B, beam_width), np.float32)
    n_hyp = np.empty((B,), np.int32)
    il = None
    if input_length is not None:
        il = np.ascontiguousarray(input_length.cpu().numpy() if torch.is_tensor(input_length) else input_length, np.int32)
    ilp = il.ctypes.data_as(ctypes.c_void_p) if il is not None else ctypes.c_void_p()
    outs = [a.ctypes.data_as(ctypes.c_void_p) for a in (ids, lens, scores, n_hyp)]
    if on_gpu:
        xd = x.to(torch.float32).contiguous()
        nbytes = ctypes.c_size_t()
        _lib.check(lib.mi355asr_ctc_prefix_beam_workspace_bytes(B, T, int(cutoff_top_n), int(beam_width), max_len, ctypes.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=xd.device)
        with torch.cuda.device(xd.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(xd.device).cuda_stream)
            _lib.check(lib.mi355asr_ctc_prefix_beam(_p(xd), int(bool(is_logits)), ilp, B, T, V, beam_width,
                                                   float(cutoff_prob), cutoff_top_n, nthreads, max_len, *outs,
                                                   _p(ws), ws.numel(), st))
    else:
        if is_logits:
            raise ValueError("host path takes probabilities (the reference's input); pass is_logits only with CUDA tensors")
        xh = np.ascontiguousarray(x.numpy() if torch.is_tensor(x) else x, np.float32)
        _lib.check(lib.mi355asr_ctc_prefix_beam_host(xh.ctypes.data_as(ctypes.c_void_p), ilp, B, T, V, beam_width,
                                                    float(cutoff_prob), cutoff_top_n, nthreads, max_len, *outs))
    return ids, lens, scores, n_hyp


class ChunkBeamPipeline:
    """ChunkConformer `predict` of batch n + 1 overlapped with the prefix beam search of batch n (round 3).

    The device beam search is a latency chain -- one workgroup per utterance walking T_pick dependent frames, 16 of the
    256 CUs busy for milliseconds -- and `predict` is a run of short chip-wide kernels, so the two share the GPU well:
    the search runs on a second HIP stream from a helper thread (ctypes releases the GIL) behind an event recorded after
    `predict`, while the caller's stream already recognises the next batch.  `push(wav)` returns the beams of the PREVIOUS
    batch (None for the first), `flush()` the last one's: (ids, lens, scores, n_hyp) as `ctc_prefix_beam_decode`."""

    def __init__(self, model, beam_width=10, cutoff_prob=0.99, cutoff_top_n=40):
        from concurrent.futures import ThreadPoolExecutor
        self.model, self.kw = model, dict(beam_width=beam_width, cutoff_prob=cutoff_prob, cutoff_top_n=cutoff_top_n)
        self.device = model._h.device
        self.side = torch.cuda.Stream(device=self.device)
        self.pool = ThreadPoolExecutor(max_workers=1)
        self.pending = None

    def _decode(self, logits, counts, ready):
        with torch.cuda.device(self.device), torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            return ctc_prefix_beam_decode(logits, counts, is_logits=True, **self.kw)

    def push(self, wav):
        logits, counts = self.model.predict(wav)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        logits.record_stream(self.side)                    # the caching allocator must not hand the logits out again early
        # the new batch's search is queued BEFORE the previous result is looked at: if that search raised, the exception
        # surfaces here once, and the pipeline goes on with the batch just predicted instead of re-raising a stale future
        old, self.pending = self.pending, self.pool.submit(self._decode, logits, counts, ready)
        return old.result() if old is not None else None

    def flush(self):
        old, self.pending = self.pending, None
        return old.result() if old is not None else None

    def close(self):
        try:
            self.flush()
        finally:
            self.pool.shutdown()


class BeamDecoder:
    """Stateful prefix beam search: externals/ctc_decoders `BeamDecoder(vocabulary, beam_size, cutoff_prob,
    cutoff_top_n)` with `.decode(probs_seq)` / `.reset()` (ctc_beam_search_decoder.cpp:217-405, no external scorer).
    `vocabulary` includes the blank as its LAST entry, as the reference class expects.  decode() returns the current
    beam as [(log_prob, text)], best first, after consuming the given frames; `decode_ids` returns token ids."""

    def __init__(self, vocabulary, beam_size, cutoff_prob=1.0, cutoff_top_n=40, ext_scorer=None):
        if ext_scorer is not None:
            raise NotImplementedError("external scorer (KenLM / OpenFST) is not part of this build")
        self.vocabulary = list(vocabulary)
        self.beam_size = int(beam_size)
        self.lib = _lib.lib()
        self.ptr = ctypes.c_void_p()
        _lib.check(self.lib.mi355asr_beam_create(len(self.vocabulary), self.beam_size, float(cutoff_prob),
                                                 int(cutoff_top_n), ctypes.byref(self.ptr)))

    def __del__(self):
        try:
            if self.ptr:
                self.lib.mi355asr_beam_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass

    def reset(self):
        _lib.check(self.lib.mi355asr_beam_reset(self.ptr))
        self._frames = 0

    def decode_ids(self, probs_seq, max_len=None):
        x = probs_seq.detach().cpu().numpy() if torch.is_tensor(probs_seq) else np.asarray(probs_seq)
        x = np.ascontiguousarray(x, np.float32).reshape(-1, len(self.vocabulary))
        self._frames = getattr(self, "_frames", 0) + x.shape[0]
        max_len = int(max_len or max(self._frames, 1))
        ids = np.empty((self.beam_size, max_len), np.int32)
        lens = np.empty((self.beam_size,), np.int32)
        scores = np.empty((self.beam_size,), np.float32)
        n = ctypes.c_int32()
        _lib.check(self.lib.mi355asr_beam_decode(self.ptr, x.ctypes.data_as(ctypes.c_void_p), x.shape[0], max_len,
                                                 ids.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p),
                                                 scores.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)))
        return [(float(scores[i]), ids[i, :lens[i]].tolist()) for i in range(n.value)]

    def decode(self, probs_seq):
        return [(sc, "".join(self.vocabulary[t] for t in toks)) for sc, toks in self.decode_ids(probs_seq)]


def _chunk_block_shapes(p, d, H, hs, k):
    s = _block_shapes(p, d, H, hs, k)
    m = p + "/mhsa_module/mha"
    for nm in ("query_kernel", "key_kernel", "value_kernel", "projection_kernel", "projection_bias"):
        del s[m + "/" + nm]
    for nm in ("query", "key", "value"):
        s[m + "/" + nm + "/kernel"] = (d, H, hs)
        s[m + "/" + nm + "/bias"] = (H, hs)
    s[m + "/attention_output/kernel"] = (H, hs, d)
    s[m + "/attention_output/bias"] = (d,)
    return s


class ChunkConformer(_ModelBase):
    """asr/models/chunk_conformer_blocks.py:775-822, offline `predict`: front -> ChunkConformerEncoder -> phone
    picker -> feature_pick -> ContextHelper -> text decoder.  `config` is the reference's model YAML as a dict
    (asr/configs/chunk_conformerS.yml); `phone` / `txt` are the two vocabularies' num_classes."""

    def __init__(self, config, phone, txt, device="cuda:0", name="chunk_conformer"):
        mc = config["model_config"]
        fr, en = mc["ChunkConformerFront"], mc["ChunkConformerEncoder"]
        pk, dc, hp = mc["ChunkCTCPicker"], mc["ChunkCTCDecoder"], mc["ContextHelper"]
        for sub in (pk, dc, hp):
            for key in ("dmodel", "head_size", "num_heads", "kernel_size"):
                if sub[key] != en[key]:
                    raise NotImplementedError("sub-models with different %s are not supported" % key)
        if en.get("padding", "causal") != "causal":
            raise NotImplementedError("ChunkConformerEncoder padding=%r: only 'causal'" % en.get("padding"))
        self.name = name
        self.dmodel, self.num_heads, self.head_size, self.kernel_size = en["dmodel"], en["num_heads"], en["head_size"], en["kernel_size"]
        self.sample_rate, self.n_mels = fr["sample_rate"], fr["n_mels"]
        self.phone_num_classes, self.txt_num_classes = phone, txt
        self.blocks = {"encoder": en["num_blocks"], "picker": pk["num_blocks"], "helper": hp["num_blocks"], "decoder": dc["num_blocks"]}
        self._config = config
        self._weights = None
        cfg = _lib.ChunkConfig(
            dmodel=en["dmodel"], head_size=en["head_size"], num_heads=en["num_heads"], kernel_size=en["kernel_size"],
            fc_factor=en["fc_factor"], n_mels=fr["n_mels"], sample_rate=fr["sample_rate"], stride_ms=fr["stride_ms"],
            n_dft=1024, reduction_factor=fr["reduction_factor"],
            enc_num_blocks=en["num_blocks"], enc_win_front=en["win_front"], enc_win_back=en["win_back"],
            picker_num_classes=phone, picker_num_blocks=pk["num_blocks"], picker_win_front=pk["win_front"], picker_win_back=pk["win_back"],
            helper_num_blocks=hp["num_blocks"], helper_win_front=hp["win_front"], helper_win_back=hp["win_back"],
            decoder_num_classes=txt, decoder_num_blocks=dc["num_blocks"], decoder_win_front=dc["win_front"], decoder_win_back=dc["win_back"])
        self._h = _Handle(cfg, device)

    def _expected_shapes(self):
        d, H, hs, k = self.dmodel, self.num_heads, self.head_size, self.kernel_size
        f2 = ((self.n_mels + 4 - 3) // 2 + 1 - 3) // 2 + 1
        s = {"front/mel_layer/real_kernels": (1024, 1, 1, 513), "front/mel_layer/imag_kernels": (1024, 1, 1, 513),
             "front/mel_layer/freq2mel": (513, self.n_mels),
             "front/conv_subsampling/conv1/kernel": (3, 3, 1, d), "front/conv_subsampling/conv1/bias": (d,),
             "front/conv_subsampling/conv2/kernel": (3, 3, d, d), "front/conv_subsampling/conv2/bias": (d,),
             "front/conv_subsampling/linear/kernel": (f2 * d, d), "front/conv_subsampling/linear/bias": (d,)}
        for i in range(self.blocks["encoder"]):
            s.update(_chunk_block_shapes("encoder/chunk_conformer_block_%d" % i, d, H, hs, k))
        for prefix, V in (("picker", self.phone_num_classes), ("helper", 0), ("decoder", self.txt_num_classes)):
            if prefix != "helper":
                s[prefix + "/project/kernel"] = (d, d)
                s[prefix + "/project/bias"] = (d,)
            for i in range(self.blocks[prefix]):
                s.update(_chunk_block_shapes("%s/block_%d" % (prefix, i), d, H, hs, k))
            if V:
                s[prefix + "/fully_connected/kernel"] = (d, V)
                s[prefix + "/fully_connected/bias"] = (V,)
        return s

    def out_frames(self, L):
        f, t = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self._h.lib.mi355asr_chunk_out_frames(self._h.ptr, L, ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def predict(self, x, stages=False):
        """x [B, L(,1)] -> text logits torch [B, T_pick, txt_num_classes] (+ counts).  With stages=True returns a dict
        with every intermediate the reference's predict() produces."""
        h = self._h
        if not h.built:
            self._build()
        xd = _wave2d(h, x)
        B, L = xd.shape
        _, T = self.out_frames(L)
        d, V = self.dmodel, self.txt_num_classes
        n = ctypes.c_size_t()
        _lib.check(h.lib.mi355asr_chunk_workspace_bytes(h.ptr, B, L, ctypes.byref(n)))
        ws = h.workspace(n.value)
        dev = h.device
        bufs = {"text_logits": torch.empty((B, T, V), dtype=torch.float32, device=dev)}
        if stages:                           # logits alone let the class head split its columns over workgroups
            bufs["text_argmax"] = torch.empty((B, T), dtype=torch.int32, device=dev)
            for k in ("front_out", "enc_out", "picker_hidden", "picked", "helper_out"):
                bufs[k] = torch.empty((B, T, d), dtype=torch.float32, device=dev)
            bufs["picker_logits"] = torch.empty((B, T, self.phone_num_classes), dtype=torch.float32, device=dev)
        outs = _lib.ChunkOutputs(**{k: v.data_ptr() for k, v in bufs.items()})
        counts = np.zeros(B, np.int32)
        tp = ctypes.c_int32()
        with torch.cuda.device(dev):
            _lib.check(h.lib.mi355asr_chunk_predict(h.ptr, _p(xd), B, L, ctypes.byref(outs),
                                                   counts.ctypes.data_as(ctypes.c_void_p), ctypes.byref(tp),
                                                   _p(ws), n.value, h._stream()))
        Tp = tp.value
        logits = bufs["text_logits"].view(-1)[:B * Tp * V].view(B, Tp, V)
        if not stages:
            return logits, counts
        r = {"text_logits": logits, "counts": counts,
             "text_argmax": bufs["text_argmax"].view(-1)[:B * Tp].view(B, Tp),
             "front": bufs["front_out"], "enc": bufs["enc_out"], "picker_logits": bufs["picker_logits"],
             "picker_hidden": bufs["picker_hidden"],
             "picked": bufs["picked"].view(-1)[:B * Tp * d].view(B, Tp, d),
             "helper": bufs["helper_out"].view(-1)[:B * Tp * d].view(B, Tp, d)}
        return r

    __call__ = predict

    # ---- streaming with explicit caches (chunk_conformer_blocks.py:799-866; single stream, B = 1) -----------------
    def _stream_cfg(self):
        mc = self._config["model_config"]
        fr = mc["ChunkConformerFront"]
        chunk_num = int(fr.get("chunk_num", 16))
        hop = int(fr["stride_ms"] * fr["sample_rate"] // 1000)
        win = {k: (mc[n]["win_front"], mc[n]["win_back"]) for k, n in (("encoder", "ChunkConformerEncoder"),
               ("picker", "ChunkCTCPicker"), ("helper", "ContextHelper"), ("decoder", "ChunkCTCDecoder"))}
        return chunk_num, hop, chunk_num // fr["reduction_factor"], win

    def init_picker_caches(self, B=1):
        """(front_wav_cache, front_sub_cache, encoder_mha_cache, encoder_cnn_cache, picker_mha_cache,
        picker_cnn_cache, dec_inp) as ChunkConformer.init_picker_caches (:799-808)."""
        if B != 1:
            raise NotImplementedError("the streaming entry points are single-stream, as the reference's (dec_inp is [1, 0, d])")
        dev, d = self._h.device, self.dmodel
        _, _, sub_length, _ = self._stream_cfg()
        z = lambda n: torch.zeros((n, 1, 0, d), dtype=torch.float32, device=dev)
        return (torch.zeros((1, 0, 1), device=dev), torch.zeros((1, sub_length, self.n_mels, 1), device=dev),
                z(self.blocks["encoder"]), z(self.blocks["encoder"]), z(self.blocks["picker"]), z(self.blocks["picker"]),
                torch.zeros((1, 0, d), device=dev))

    def init_decoder_caches(self, B=1):
        """(helper_mha_cache, helper_cnn_cache, decoder_mha_cache, decoder_cnn_cache, dec_inp) (:810-814)."""
        if B != 1:
            raise NotImplementedError("the streaming entry points are single-stream, as the reference's")
        dev, d = self._h.device, self.dmodel
        z = lambda n: torch.zeros((n, 1, 0, d), dtype=torch.float32, device=dev)
        return (z(self.blocks["helper"]), z(self.blocks["helper"]), z(self.blocks["decoder"]), z(self.blocks["decoder"]),
                torch.zeros((1, 0, d), device=dev))

    def _stream_ws(self, max_rows, Lw, S, chunk_num):
        h = self._h
        n = ctypes.c_size_t()
        _lib.check(h.lib.mi355asr_chunk_stream_workspace_bytes(h.ptr, int(max_rows), int(max(Lw, 1)), int(S), int(chunk_num),
                                                               ctypes.byref(n)))
        return h.workspace(n.value), n.value

    def _stack_stream(self, stack, x, mha_caches, cnn_caches, want_logits):
        """one *.stream_call up to the valid/unvalid slicing: x [1,T,d], caches [nblk,1,C,d] -> hidden [1,T,d],
        logits [1,T,V] or None, new (untrimmed) caches [nblk,1,C+T,d]."""
        h = self._h
        nblk, _, Cm, d = mha_caches.shape
        Cc = cnn_caches.shape[2]
        T = x.shape[1]
        dev = h.device
        x = x.contiguous()
        mha_caches, cnn_caches = mha_caches.contiguous(), cnn_caches.contiguous()
        hidden = torch.empty((1, T, d), dtype=torch.float32, device=dev)
        V = {1: self.phone_num_classes, 3: self.txt_num_classes}.get(stack)
        logits = torch.empty((1, T, V), dtype=torch.float32, device=dev) if (want_logits and V) else None
        new_mha = torch.empty((nblk, 1, Cm + T, d), dtype=torch.float32, device=dev)
        new_cnn = torch.empty((nblk, 1, Cc + T, d), dtype=torch.float32, device=dev)
        ws, n = self._stream_ws(max(Cm, Cc) + T, 1, 0, 16)
        with torch.cuda.device(dev):
            _lib.check(h.lib.mi355asr_chunk_stack_stream(h.ptr, stack, _p(x), T, _p(mha_caches) if Cm else None, Cm,
                                                         _p(cnn_caches) if Cc else None, Cc, _p(hidden), _p(logits), None,
                                                         _p(new_mha), _p(new_cnn), _p(ws), n, h._stream()))
        return hidden, logits, new_mha, new_cnn

    def _stack_stream_call(self, name, stack, x, mha_caches, cnn_caches, head):
        """ChunkConformerEncoder / ChunkCTCDecoder / ContextHelper .stream_call including the slicing (:546-558,
        660-672, 764-770) -> (valid_logits, valid_hidden, new_mha, new_cnn, unvalid_logits)."""
        _, _, _, win = self._stream_cfg()
        wf, wb = win[name]
        k = self.kernel_size
        hidden, logits, new_mha, new_cnn = self._stack_stream(stack, x, mha_caches, cnn_caches, head)
        if wb != 0:
            cut = lambda a, ax: a.narrow(ax, 0, max(a.shape[ax] - wb, 0))
            valid_hidden = cut(hidden, 1)
            valid_logits = cut(logits, 1) if head else None
            unvalid = logits[:, -wb:] if head else None
            new_mha, new_cnn = cut(new_mha, 2), cut(new_cnn, 2)
        else:
            valid_hidden, valid_logits = hidden, logits
            unvalid = torch.zeros_like(logits) if head else None
        return valid_logits, valid_hidden, new_mha[:, :, -wf:], new_cnn[:, :, -k:], unvalid

    def picker_stream_predict(self, input_wav, caches):
        """ChunkConformer.picker_stream_predict (:824-842): input_wav [1, L, 1] (the reference feeds chunk_num * hop
        = 2560 samples per call) -> (valid_ctc_out, unvalid_ctc_out, valid_hidden_out, caches)."""
        h = self._h
        if not h.built:
            self._build()
        front_wav_cache, front_sub_cache, enc_mha, enc_cnn, pk_mha, pk_cnn, dec_inp = caches
        chunk_num, hop, sub_length, _ = self._stream_cfg()
        dev, d = h.device, self.dmodel
        wav = _wave2d(h, input_wav)
        if wav.shape[0] != 1:
            raise NotImplementedError("single stream only")
        new_wav = torch.cat([front_wav_cache.reshape(1, -1), wav], 1).contiguous()          # :449
        Lw = new_wav.shape[1]
        sub = front_sub_cache.reshape(1, -1, self.n_mels).contiguous()
        S = sub.shape[1]
        nf, tout = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(h.lib.mi355asr_chunk_front_stream_shape(h.ptr, Lw, S, chunk_num, ctypes.byref(nf), ctypes.byref(tout)))
        front = torch.empty((1, tout.value, d), dtype=torch.float32, device=dev)
        new_sub = torch.empty((1, S + nf.value, self.n_mels), dtype=torch.float32, device=dev)
        ws, n = self._stream_ws(max(tout.value, 1), Lw, S, chunk_num)
        with torch.cuda.device(dev):
            _lib.check(h.lib.mi355asr_chunk_front_stream(h.ptr, _p(new_wav), Lw, _p(sub) if S else None, S, chunk_num,
                                                         _p(front), _p(new_sub), _p(ws), n, h._stream()))
        new_wav_cache = new_wav[:, -chunk_num * hop:].reshape(1, -1, 1)                       # :456
        new_sub_cache = new_sub[:, -sub_length:].reshape(1, -1, self.n_mels, 1)               # :457
        if tout.value == 0:
            empty = torch.zeros((1, 0, self.phone_num_classes), device=dev)
            return empty, empty, torch.zeros((1, 0, d), device=dev), (new_wav_cache, new_sub_cache, enc_mha, enc_cnn,
                                                                    pk_mha, pk_cnn, dec_inp)
        _, valid_enc, enc_mha, enc_cnn, _ = self._stack_stream_call("encoder", 0, front, enc_mha, enc_cnn, False)
        dec_inp = torch.cat([dec_inp, valid_enc], 1)                                          # :830
        if dec_inp.shape[1] == 0:
            empty = torch.zeros((1, 0, self.phone_num_classes), device=dev)
            return empty, empty, dec_inp, (new_wav_cache, new_sub_cache, enc_mha, enc_cnn, pk_mha, pk_cnn, dec_inp)
        valid_ctc, valid_hidden, pk_mha, pk_cnn, unvalid_ctc = self._stack_stream_call("picker", 1, dec_inp, pk_mha, pk_cnn, True)
        dec_inp = dec_inp[:, valid_ctc.shape[1]:]                                             # :833-834
        return valid_ctc, unvalid_ctc, valid_hidden, (new_wav_cache, new_sub_cache, enc_mha, enc_cnn, pk_mha, pk_cnn, dec_inp)

    def decoder_stream_predict(self, valid_enc_out, caches):
        """ChunkConformer.decoder_stream_predict (:844-857): picked features [1, Tp, d] -> (valid_ctc_out,
        unvalid_ctc_out, caches)."""
        h = self._h
        helper_mha, helper_cnn, dec_mha, dec_cnn, dec_inp = caches
        x = h.to_device(valid_enc_out)
        _, helped, helper_mha, helper_cnn, _ = self._stack_stream_call("helper", 2, x, helper_mha, helper_cnn, False)
        dec_inp = torch.cat([dec_inp, helped], 1)
        valid_ctc, _, dec_mha, dec_cnn, unvalid_ctc = self._stack_stream_call("decoder", 3, dec_inp, dec_mha, dec_cnn, True)
        dec_inp = dec_inp[:, valid_ctc.shape[1]:]
        return valid_ctc, unvalid_ctc, (helper_mha, helper_cnn, dec_mha, dec_cnn, dec_inp)

    def feature_pick(self, encoder_hidden_states, ctc_outs, max_T=None):
        """ChunkConformer.feature_pick (:913-999): keep the frames whose argmax is not the blank, compacted per
        utterance, zero padded to the batch maximum -> (feature_outputs [B,Tp,d], ctc_outputs [B,Tp,V]).  Argmax,
        compaction and gather run in libmi355asr.so (`mi355asr_feature_pick_count` / `_gather`); the only host step is
        reading the B counts that size the outputs."""
        h = self._h
        hid = h.to_device(encoder_hidden_states)
        ctc = h.to_device(ctc_outs)
        B, T, d = hid.shape
        V = ctc.shape[-1]
        if ctc.shape[0] != B or ctc.shape[1] != T:
            raise ValueError("hidden %s and ctc %s disagree" % (tuple(hid.shape), tuple(ctc.shape)))
        if V != self.phone_num_classes:
            raise ValueError("ctc_outs has %d classes, the picker %d" % (V, self.phone_num_classes))
        idx = torch.empty((B, max(T, 1)), dtype=torch.int32, device=h.device)
        cnt = torch.empty((B,), dtype=torch.int32, device=h.device)
        counts = np.zeros(B, np.int32)
        with torch.cuda.device(h.device):
            _lib.check(h.lib.mi355asr_feature_pick_count(_p(ctc), B, T, V, _p(idx), _p(cnt),
                                                         counts.ctypes.data_as(ctypes.c_void_p), h._stream()))
            Tp = int(counts.max()) if B else 0
            if max_T is not None:
                Tp = max(Tp, int(max_T))
            f = torch.empty((B, Tp, d), dtype=torch.float32, device=h.device)
            c = torch.empty((B, Tp, V), dtype=torch.float32, device=h.device)
            if Tp == 0:
                return f, c
            _lib.check(h.lib.mi355asr_feature_pick_gather(_p(hid), _p(ctc), _p(idx), _p(cnt), B, T, d, V, Tp, _p(f), _p(c),
                                                          h._stream()))
        return f, c
