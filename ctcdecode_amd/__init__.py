"""MI355X-native drop-in for ``ctcdecode.CTCBeamDecoder.decode()`` (no language model).

Same constructor arguments, same ``decode(probs, seq_lens)`` call and the same four output tensors as
ctcdecode/__init__.py:6-123 of the reference; the prefix beam search itself runs as a hand-written HIP kernel
(one workgroup per utterance, beam in LDS) behind the C ABI in include/ctcdecode_amd.h.
"""
import ctypes
import weakref

import torch

from . import _native
from ._native import NativeError  # noqa: F401

__all__ = ["CTCBeamDecoder", "OnlineCTCBeamDecoder", "DecoderState", "NativeError", "CallbackScorer", "KenlmScorer", "DecodePipeline"]
HAVE_LM = True  # the external-scorer tier (model_path / alpha / beta) is part of this build


class _Scorer(object):
    """ctypes owner of a ``ctcd_scorer`` (paddle_get_scorer / paddle_release_scorer, ctcdecode/__init__.py:47-50,138-140)."""

    def __init__(self, alpha, beta, model_path, labels, device_index):
        if isinstance(model_path, bytes):
            model_path = model_path.decode("utf-8")
        arr = (ctypes.c_char_p * len(labels))(*[str(x).encode("utf-8") for x in labels])
        h = ctypes.c_void_p()
        _native.check(_native.lib.ctcd_scorer_create(ctypes.byref(h), float(alpha), float(beta), str(model_path).encode("utf-8"), arr,
                                                     len(labels), int(device_index)))
        self.handle = h

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                _native.lib.ctcd_scorer_destroy(h)
            except Exception:
                pass
            self.handle = None


class CallbackScorer(object):
    """The swappable scorer: a language model behind a host callback (``ctcd_scorer_create_callback``,
    include/ctcdecode_amd.h) -- the counterpart of handing the reference's decoder another ``Scorer`` implementation
    (ctcdecode/src/scorer.h:41-78).  Pass it to ``CTCBeamDecoder(..., scorer=...)`` / ``OnlineCTCBeamDecoder``.

    ``cond_log10(words)`` receives the window's ``max_order`` words (tuple of str, oldest first, ``"<s>"``-padded) and
    returns log10 p(words[-1] | words[:-1]) -- or ``None`` when the window holds a word the model does not know
    (the reference's OOV_SCORE, scorer.cpp:86-88).  It must be a pure function of the words: every distinct window is
    asked for once and cached on the device.  ``vocabulary``: the model's words (builds the dictionary of a word model;
    only single characters = character model, scorer.cpp:65-71).  An exception inside the callback fails the decode and
    is re-raised from it.
    """

    def __init__(self, cond_log10, vocabulary, max_order, labels, alpha=0.0, beta=0.0, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("ctcdecode_amd: no HIP device visible; this decoder has no CPU path")
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        self._fn = cond_log10
        self._error = None

        def trampoline(_user, words, n, out):
            try:
                r = self._fn(tuple(words[i].decode("utf-8") for i in range(n)))
                if r is None:
                    return 1
                out[0] = float(r)
                return 0
            except BaseException as e:  # (an exception must not unwind through the C frames)
                self._error = e
                return -1

        self._c_fn = _native.COND_LOG10_FN(trampoline)  # kept alive with the scorer
        voc = [str(w).encode("utf-8") for w in vocabulary]
        varr = (ctypes.c_char_p * max(len(voc), 1))(*voc)
        larr = (ctypes.c_char_p * len(labels))(*[str(x).encode("utf-8") for x in labels])
        h = ctypes.c_void_p()
        _native.check(_native.lib.ctcd_scorer_create_callback(ctypes.byref(h), float(alpha), float(beta), int(max_order), varr, len(voc), self._c_fn, None,
                                                              larr, len(labels), int(index)))
        self.handle = h
        self.device_index = int(index)
        self.num_labels = len(labels)

    @classmethod
    def from_c(cls, fn_address, user_address, vocabulary, max_order, labels, alpha=0.0, beta=0.0, device=None, keepalive=None):
        """The hook with a NATIVE callback: ``fn_address`` = address of a C function of type ``ctcd_cond_log10_fn``
        (include/ctcdecode_amd.h: ``int fn(void *user, const char *const *words, int n, float *log10_prob)``), ``user_address``
        its first argument -- e.g. a thin shim over kenlm's C++ API: no Python between the decoder and the model.
        (``ctcd_scorer_cond_log10`` has this very signature: any built-in scorer can sit behind the hook -- bench.py does that to
        price the hook itself.)  ``keepalive``: objects that must outlive the scorer."""
        self = cls.__new__(cls)
        if not torch.cuda.is_available():
            raise RuntimeError("ctcdecode_amd: no HIP device visible; this decoder has no CPU path")
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        self._fn, self._error, self._keepalive = None, None, keepalive
        self._c_fn = ctypes.cast(ctypes.c_void_p(int(fn_address)), _native.COND_LOG10_FN)
        voc = [str(w).encode("utf-8") for w in vocabulary]
        varr = (ctypes.c_char_p * max(len(voc), 1))(*voc)
        larr = (ctypes.c_char_p * len(labels))(*[str(x).encode("utf-8") for x in labels])
        h = ctypes.c_void_p()
        _native.check(_native.lib.ctcd_scorer_create_callback(ctypes.byref(h), float(alpha), float(beta), int(max_order), varr, len(voc), self._c_fn,
                                                              ctypes.c_void_p(int(user_address)) if user_address else None, larr, len(labels), int(index)))
        self.handle = h
        self.device_index = int(index)
        self.num_labels = len(labels)
        return self

    def callback_calls(self):
        """Distinct windows the callback has been asked for so far."""
        return int(_native.lib.ctcd_scorer_callback_calls(self.handle))

    def callback_seconds(self):
        """Time spent inside the callback so far (an estimate: every 16th call is timed)."""
        return float(_native.lib.ctcd_scorer_callback_seconds(self.handle))

    def set_callback_threads(self, threads):
        """A NATIVE callback that may be called from several threads at once (``from_c`` over a read-only model): ``threads - 1``
        helper threads ask beside the calling one while a launch waits for its answers (``ctcd_scorer_set_callback_threads``).
        Refused for Python callables: the interpreter lock would serialise them."""
        if self._fn is not None and int(threads) != 1:
            raise ValueError("callback threads are for native callbacks (CallbackScorer.from_c): a Python callable runs under the interpreter lock")
        _native.check(_native.lib.ctcd_scorer_set_callback_threads(self.handle, int(threads)))

    def _raise_pending(self):
        e, self._error = self._error, None
        if e is not None:
            raise e

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                _native.lib.ctcd_scorer_destroy(h)
            except Exception:
                pass
            self.handle = None


class KenlmScorer(CallbackScorer):
    """A kenlm model -- binary files included -- behind the scorer hook, through the ``kenlm`` Python module (not part of
    this image: ImportError if it is missing).  ``vocabulary``: the model's words (a binary model does not list them; pass
    the word list it was trained with, as the reference's Scorer reads them from the model, scorer.cpp:196-230)."""

    def __init__(self, model_path, vocabulary, labels, alpha=0.0, beta=0.0, device=None):
        import kenlm  # noqa: F401  (optional dependency)

        self._model = kenlm.Model(str(model_path))
        model = self._model

        def cond_log10(words):
            # Scorer::get_log_cond_prob (scorer.cpp:74-93): feed the window from the empty context, OOV if any word is unknown
            state, out = kenlm.State(), kenlm.State()
            model.NullContextWrite(state)
            p = 0.0
            for w in words:
                if w not in model:
                    return None
                p = model.BaseScore(state, w, out)
                state, out = out, state
            return p

        CallbackScorer.__init__(self, cond_log10, vocabulary, model.order, labels, alpha, beta, device)


def _to_host(tensors):
    """Device -> host copy of the result tensors through page-locked memory (PyTorch's caching host allocator recycles
    the blocks), all copies in flight together, one synchronisation: several times faster than ``.cpu()`` on pageable
    memory for the 2 x [B, K, T] int32 results."""
    outs = []
    for t in tensors:
        h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
        h.copy_(t, non_blocking=True)
        outs.append(h)
    torch.cuda.current_stream(tensors[0].device).synchronize()
    return tuple(outs)


def _adopt_scorer(scorer, model_path, num_labels, device_index):
    if model_path:
        raise ValueError("pass either model_path or scorer, not both")
    if not isinstance(scorer, CallbackScorer):
        raise TypeError("scorer must be a ctcdecode_amd.CallbackScorer (or KenlmScorer)")
    if scorer.num_labels != num_labels or scorer.device_index != device_index:
        raise ValueError("the scorer was built for other labels / another device than this decoder")
    return scorer


class CTCBeamDecoder(object):
    """See ctcdecode/__init__.py:6-51 of the reference for the meaning of the arguments.

    Differences, all additive:
      * ``device``: the MI355X to decode on (default: ``cuda:<current>``).
      * ``decode_device()`` returns the four tensors in HBM without the device->host copy.
      * positions the reference leaves uninitialised (``[b, p, out_len:]``, rows ``p >= #results``) are zero.
      * ``model_path`` must be an ARPA text model (binary kenlm files are not supported); the scorer's tables are mirrored
        into HBM and queried inside the decode kernel (include/ctcdecode_amd.h, "LM tier").
    """

    def __init__(self, labels, model_path=None, alpha=0, beta=0, cutoff_top_n=40, cutoff_prob=1.0, beam_width=100,
                 num_processes=4, blank_id=0, log_probs_input=False, device=None, logits_input=False, scorer=None):
        self.cutoff_top_n = cutoff_top_n
        self._beam_width = beam_width
        self._scorer = None
        self._num_processes = num_processes
        self._labels = list(labels)
        self._num_labels = len(labels)
        self._blank_id = blank_id
        # logits_input (extension, not in the reference): the input holds raw logits; the device normalises them with a
        # float32 log_softmax (include/ctcdecode_amd.h ctcd_log_softmax) before decoding
        self._log_probs = 2 if logits_input else (1 if log_probs_input else 0)
        self._cutoff_prob = cutoff_prob
        if not torch.cuda.is_available():
            raise RuntimeError("ctcdecode_amd: no HIP device visible; this decoder has no CPU path")
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self._device.type != "cuda":
            raise ValueError("ctcdecode_amd: device must be a HIP (cuda:N) device")
        if self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        h = ctypes.c_void_p()
        _native.check(_native.lib.ctcd_create(ctypes.byref(h), self._device.index))
        self._handle = h
        if scorer is not None:  # a ready-made scorer object (CallbackScorer / KenlmScorer): the swappable-scorer hook
            self._scorer = _adopt_scorer(scorer, model_path, self._num_labels, self._device.index)
        elif model_path:  # ctcdecode/__init__.py:47-50 (`if model_path:`: None and "" both mean no scorer)
            self._scorer = _Scorer(alpha, beta, model_path, self._labels, self._device.index)

    def _check(self, rc):
        """``_native.check`` for the calls a scorer callback runs under: an exception raised inside the callback comes first."""
        if rc and self._scorer is not None and hasattr(self._scorer, "_raise_pending"):
            self._scorer._raise_pending()
        _native.check(rc)

    def set_threads(self, n):
        _native.check(_native.lib.ctcd_set_threads(self._handle, int(n)))

    def set_cu_sharing(self, mode=1):
        """1: always launch the two-workgroups-per-CU build of the kernel (beam <= 128, <= 32 labels) -- for a serving loop
        that keeps several launches in flight; 0: never; -1 (default): when a batch has more utterances than the GPU has CUs."""
        _native.check(_native.lib.ctcd_set_cu_sharing(self._handle, int(mode)))

    def set_subtree_search(self, mode=-1):
        """Phase A1's four-subtrees-per-wave build of the north-star class kernel (include/ctcdecode_amd.h): -1 = chosen from the beam
        shape the last checked launch reported (chain-shaped beams, i.e. blank-dominated rows: on), 0 / 1 = never / always."""
        _native.check(_native.lib.ctcd_set_subtree_search(self._handle, int(mode)))

    def last_subtree_search(self):
        return int(_native.lib.ctcd_last_subtree_search(self._handle))

    def set_host_path(self, input_streaming=None, mirror_cap_labels=None):
        """Test hook for decode(): turn the streamed input off / on; shrink the host mirror of the compact results."""
        _native.check(_native.lib.ctcd_debug_set_host_path(self._handle, -1 if input_streaming is None else int(bool(input_streaming)),
                                                           -2 if mirror_cap_labels is None else int(mirror_cap_labels)))

    def set_fixed_layout(self, on=True):
        """Test hook: False forces the run-time workspace layout also for small shapes (identical results)."""
        _native.check(_native.lib.ctcd_debug_set_fixed_layout(self._handle, 1 if on else 0))

    def set_scorer_wait(self, on=True):
        """Callback scorers: True (default) = a launch waits on the GPU for the callback's answers; False = every miss ends the
        utterance's launch and the host relaunches (the form of rounds 4-5).  Identical results."""
        _native.check(_native.lib.ctcd_set_scorer_wait(self._handle, 1 if on else 0))

    def last_scorer_launches(self):
        """(launches, answer batches handed to waiting launches) of the last decode with a callback scorer."""
        return int(_native.lib.ctcd_last_scorer_rounds(self._handle)), int(_native.lib.ctcd_last_scorer_waits(self._handle))

    def set_fused_logits(self, on=True):
        """Test hook (logits_input=True): False sends raw logits through the one-wave log_softmax pass and the separate prune
        instead of the workgroup kernels / the fused logits-to-candidates pass (identical results)."""
        _native.check(_native.lib.ctcd_debug_set_fused_logits(self._handle, 1 if on else 0))

    def set_prune_registers(self, on=True):
        """Test hook: False makes the workgroup prune pass read every row twice (the form of rounds 2-5) instead of keeping it in
        registers (identical results)."""
        _native.check(_native.lib.ctcd_debug_set_prune_registers(self._handle, 1 if on else 0))

    def last_prune_rows(self, rows, stride):
        """Test hook: the vocabulary-prune pass's output of the last call as numpy arrays (counts[rows], labels and values
        [rows, stride], stride = min(cutoff_top_n, V); entries at or beyond a frame's count are unspecified)."""
        import numpy as np

        cnt = np.zeros((rows,), np.int32)
        lab = np.zeros((rows, stride), np.int32)
        val = np.zeros((rows, stride), np.float32)
        _native.check(_native.lib.ctcd_debug_prune_rows(self._handle, rows, stride, cnt.ctypes.data, lab.ctypes.data, val.ctypes.data))
        return cnt, lab, val

    def set_timing(self, on=True):
        _native.check(_native.lib.ctcd_set_timing(self._handle, 1 if on else 0))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        _native.check(_native.lib.ctcd_last_kernel_ms(self._handle, ctypes.byref(ms)))
        return float(ms.value)

    def last_prune_ms(self):
        ms = ctypes.c_float()
        _native.check(_native.lib.ctcd_last_prune_ms(self._handle, ctypes.byref(ms)))
        return float(ms.value)

    def decode_device(self, probs, seq_lens=None, check=True):
        """``probs``: [B, T, V] tensor (any device / float dtype).  Returns (beam_results, beam_scores, timesteps,
        out_lens) as tensors in HBM on the decoder's device; asynchronous on the current stream when ``check`` is False."""
        if probs.dim() != 3:
            raise ValueError("probs must be [batch, time, labels]")
        probs = probs.to(device=self._device, dtype=torch.float32).contiguous()
        B, T, V = probs.shape
        if V != self._num_labels:
            raise ValueError("probs.shape[2] (%d) does not match the number of labels (%d)" % (V, self._num_labels))
        if seq_lens is not None:
            seq_lens = seq_lens.to(device=self._device, dtype=torch.int32).contiguous()
            if seq_lens.numel() != B:
                raise ValueError("seq_lens must have one entry per batch item")
        K = self._beam_width
        with torch.cuda.device(self._device):
            output = torch.empty((B, K, T), dtype=torch.int32, device=self._device)
            timesteps = torch.empty((B, K, T), dtype=torch.int32, device=self._device)
            scores = torch.empty((B, K), dtype=torch.float32, device=self._device)
            out_len = torch.empty((B, K), dtype=torch.int32, device=self._device)
            stream = torch.cuda.current_stream(self._device).cuda_stream
            if self._scorer is not None:  # ctcdecode/__init__.py:87-104 (paddle_beam_decode_lm)
                self._check(_native.lib.ctcd_beam_decode_lm(
                    self._handle, probs.data_ptr(), seq_lens.data_ptr() if seq_lens is not None else None, B, T, V, K,
                    self._num_processes, float(self._cutoff_prob), int(self.cutoff_top_n), int(self._blank_id), self._log_probs,
                    self._scorer.handle, output.data_ptr(), timesteps.data_ptr(), scores.data_ptr(), out_len.data_ptr(), None, stream))
            else:
                _native.check(_native.lib.ctcd_beam_decode(
                    self._handle, probs.data_ptr(), seq_lens.data_ptr() if seq_lens is not None else None, B, T, V, K,
                    self._num_processes, float(self._cutoff_prob), int(self.cutoff_top_n), int(self._blank_id), self._log_probs,
                    output.data_ptr(), timesteps.data_ptr(), scores.data_ptr(), out_len.data_ptr(), None, stream))
            if check:
                _native.check(_native.lib.ctcd_check_status(self._handle, B))
        return output, scores, timesteps, out_len

    def decode(self, probs, seq_lens=None):
        """Drop-in for ctcdecode/__init__.py:53-123: returns CPU tensors (output, scores, timesteps, out_seq_len).

        The results leave the GPU in compact form (every beam only the labels it does not share with its neighbour in the
        trie, include/ctcdecode_amd.h) and ``num_processes`` host threads expand them into the four tensors."""
        if probs.dim() != 3:
            raise ValueError("probs must be [batch, time, labels]")
        on_dev = probs.is_cuda
        if on_dev:
            probs = probs.to(device=self._device, dtype=torch.float32).contiguous()
        else:
            probs = probs.to(dtype=torch.float32).contiguous()  # ctcdecode/__init__.py:77
        B, T, V = probs.shape
        if V != self._num_labels:
            raise ValueError("probs.shape[2] (%d) does not match the number of labels (%d)" % (V, self._num_labels))
        if seq_lens is not None:
            seq_lens = seq_lens.to(device=self._device if on_dev else "cpu", dtype=torch.int32).contiguous()
            if seq_lens.numel() != B:
                raise ValueError("seq_lens must have one entry per batch item")
        K = self._beam_width
        pin = B * K * T > 0
        output = torch.empty((B, K, T), dtype=torch.int32, pin_memory=pin)
        timesteps = torch.empty((B, K, T), dtype=torch.int32, pin_memory=pin)
        scores = torch.empty((B, K), dtype=torch.float32, pin_memory=pin)
        out_len = torch.empty((B, K), dtype=torch.int32, pin_memory=pin)
        with torch.cuda.device(self._device):
            stream = torch.cuda.current_stream(self._device).cuda_stream
            self._check(_native.lib.ctcd_beam_decode_to_host(
                self._handle, probs.data_ptr(), seq_lens.data_ptr() if seq_lens is not None else None, 1 if on_dev else 0, B, T, V, K,
                self._num_processes, float(self._cutoff_prob), int(self.cutoff_top_n), int(self._blank_id), self._log_probs,
                self._scorer.handle if self._scorer is not None else None, output.data_ptr(), timesteps.data_ptr(), scores.data_ptr(),
                out_len.data_ptr(), None, stream))
        return output, scores, timesteps, out_len

    def log_softmax(self, logits, seq_lens=None):
        """The float32 log_softmax that ``logits_input=True`` applies before decoding, as a tensor in HBM ([B, T, V]; frames at
        or beyond ``seq_lens`` are left 0).  Bit-reproducible: see include/ctcdecode_amd.h ctcd_log_softmax."""
        if logits.dim() != 3:
            raise ValueError("logits must be [batch, time, labels]")
        logits = logits.to(device=self._device, dtype=torch.float32).contiguous()
        B, T, V = logits.shape
        if seq_lens is not None:
            seq_lens = seq_lens.to(device=self._device, dtype=torch.int32).contiguous()
        with torch.cuda.device(self._device):
            out = torch.zeros_like(logits)
            stream = torch.cuda.current_stream(self._device).cuda_stream
            _native.check(_native.lib.ctcd_log_softmax(self._handle, logits.data_ptr(), seq_lens.data_ptr() if seq_lens is not None else None,
                                                       B, T, V, out.data_ptr(), stream))
        return out

    def decode_padded(self, probs, seq_lens=None):
        """The same call with the padded [B, K, T] tensors crossing PCIe (the delivery of round 1; kept for comparison)."""
        return _to_host(self.decode_device(probs, seq_lens))

    def decode_compact(self, probs, seq_lens=None):
        """Device-resident compact results: (c_hdr [B,4], c_ent [B,K,4], c_labels [n], scores [B,K], out_lens [B,K]) in HBM --
        what a rank ships to the gathering rank; ``expand_compact`` rebuilds the padded tensors there."""
        if probs.dim() != 3:
            raise ValueError("probs must be [batch, time, labels]")
        probs = probs.to(device=self._device, dtype=torch.float32).contiguous()
        B, T, V = probs.shape
        if V != self._num_labels:
            raise ValueError("probs.shape[2] (%d) does not match the number of labels (%d)" % (V, self._num_labels))
        if seq_lens is not None:
            seq_lens = seq_lens.to(device=self._device, dtype=torch.int32).contiguous()
        K = self._beam_width
        cap = int(_native.lib.ctcd_compact_label_capacity(B, K, T))
        with torch.cuda.device(self._device):
            if getattr(self, "_c_labels", None) is None or self._c_labels.numel() < max(cap, 1):
                self._c_labels = torch.empty((max(cap, 1),), dtype=torch.int32, device=self._device)  # worst case, reused
            hdr = torch.empty((B, 4), dtype=torch.int32, device=self._device)
            ent = torch.empty((B, K, 4), dtype=torch.int32, device=self._device)
            cnt = torch.empty((1,), dtype=torch.int32, device=self._device)
            scores = torch.empty((B, K), dtype=torch.float32, device=self._device)
            out_len = torch.empty((B, K), dtype=torch.int32, device=self._device)
            stream = torch.cuda.current_stream(self._device).cuda_stream
            self._check(_native.lib.ctcd_beam_decode_compact(
                self._handle, probs.data_ptr(), seq_lens.data_ptr() if seq_lens is not None else None, B, T, V, K, self._num_processes,
                float(self._cutoff_prob), int(self.cutoff_top_n), int(self._blank_id), self._log_probs,
                self._scorer.handle if self._scorer is not None else None, hdr.data_ptr(), ent.data_ptr(), self._c_labels.data_ptr(),
                cnt.data_ptr(), cap, scores.data_ptr(), out_len.data_ptr(), None, stream))
            _native.check(_native.lib.ctcd_check_status(self._handle, B))
            n = int(cnt.item())
        # (an owned copy: the per-decoder buffer is overwritten by this decoder's next call, possibly while an asynchronous
        #  gather of this batch still reads the labels)
        return hdr, ent, self._c_labels[:n].clone(), scores, out_len

    def decode_compact_async(self, probs, seq_lens=None):
        """``decode_compact`` without waiting: the kernel, then the copies of the status words and of the label count into
        page-locked memory, are enqueued on the current stream and an event is recorded behind them.  Returns a ticket for
        ``finish_compact``; the caller may queue the NEXT batch (on another decoder object: this one's workspace is in use)
        before looking at this one -- how a serving loop keeps the GPU busy while the host handles the previous results.
        With a callback scorer the call BLOCKS until the decode is done: the callback is served on the calling thread."""
        if probs.dim() != 3:
            raise ValueError("probs must be [batch, time, labels]")
        probs = probs.to(device=self._device, dtype=torch.float32).contiguous()
        B, T, V = probs.shape
        if V != self._num_labels:
            raise ValueError("probs.shape[2] (%d) does not match the number of labels (%d)" % (V, self._num_labels))
        if seq_lens is not None:
            seq_lens = seq_lens.to(device=self._device, dtype=torch.int32).contiguous()
        K = self._beam_width
        cap = int(_native.lib.ctcd_compact_label_capacity(B, K, T))
        with torch.cuda.device(self._device):
            if getattr(self, "_c_labels", None) is None or self._c_labels.numel() < max(cap, 1):
                self._c_labels = torch.empty((max(cap, 1),), dtype=torch.int32, device=self._device)  # worst case, reused
            hdr = torch.empty((B, 4), dtype=torch.int32, device=self._device)
            ent = torch.empty((B, K, 4), dtype=torch.int32, device=self._device)
            cnt = torch.empty((1,), dtype=torch.int32, device=self._device)
            scores = torch.empty((B, K), dtype=torch.float32, device=self._device)
            out_len = torch.empty((B, K), dtype=torch.int32, device=self._device)
            stream = torch.cuda.current_stream(self._device)
            self._check(_native.lib.ctcd_beam_decode_compact(
                self._handle, probs.data_ptr(), seq_lens.data_ptr() if seq_lens is not None else None, B, T, V, K, self._num_processes,
                float(self._cutoff_prob), int(self.cutoff_top_n), int(self._blank_id), self._log_probs,
                self._scorer.handle if self._scorer is not None else None, hdr.data_ptr(), ent.data_ptr(), self._c_labels.data_ptr(),
                cnt.data_ptr(), cap, scores.data_ptr(), out_len.data_ptr(), None, stream.cuda_stream))
            status = torch.empty((max(B, 1),), dtype=torch.int32, pin_memory=True)
            cnt_host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
            _native.check(_native.lib.ctcd_fetch_status_async(self._handle, B, status.data_ptr(), stream.cuda_stream))
            cnt_host.copy_(cnt, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        return dict(B=B, hdr=hdr, ent=ent, labels=self._c_labels, cnt=cnt, scores=scores, lens=out_len, status=status, cnt_host=cnt_host,
                    event=ev, keep=(probs, seq_lens))

    def finish_compact(self, ticket):
        """Waits for a ``decode_compact_async`` batch (its own event only, not the stream) and returns what ``decode_compact``
        returns: (c_hdr, c_ent, c_labels[:n], scores, out_lens)."""
        ticket["event"].synchronize()
        st = ticket["status"][:ticket["B"]]
        if ticket["B"] and bool((st != 0).any()):
            b = int((st != 0).nonzero()[0])
            raise _native.NativeError("ctcdecode_amd: decoder status %d for item %d" % (int(st[b]), b))
        n = int(ticket["cnt_host"][0])
        with torch.cuda.device(self._device):
            labels = ticket["labels"][:n].clone()  # owned: the decoder's label buffer belongs to its next call
        return ticket["hdr"], ticket["ent"], labels, ticket["scores"], ticket["lens"]

    def expand_compact(self, hdr, ent, labels, T):
        """(c_hdr, c_ent, c_labels) of any number of items -> (output [B,K,T], timesteps [B,K,T]) in HBM."""
        B, K = int(ent.shape[0]), int(ent.shape[1])
        with torch.cuda.device(self._device):
            output = torch.empty((B, K, T), dtype=torch.int32, device=self._device)
            timesteps = torch.empty((B, K, T), dtype=torch.int32, device=self._device)
            stream = torch.cuda.current_stream(self._device).cuda_stream
            _native.check(_native.lib.ctcd_expand_compact(self._handle, hdr.contiguous().data_ptr(), ent.contiguous().data_ptr(),
                                                          labels.contiguous().data_ptr(), B, K, T, output.data_ptr(), timesteps.data_ptr(), stream))
        return output, timesteps

    def character_based(self):  # ctcdecode/__init__.py:125-136: None without a scorer
        return bool(_native.lib.ctcd_scorer_is_character_based(self._scorer.handle)) if self._scorer else None

    def max_order(self):
        return int(_native.lib.ctcd_scorer_max_order(self._scorer.handle)) if self._scorer else None

    def dict_size(self):
        return int(_native.lib.ctcd_scorer_dict_size(self._scorer.handle)) if self._scorer else None

    def reset_params(self, alpha, beta):
        if self._scorer is not None:
            _native.check(_native.lib.ctcd_scorer_reset_params(self._scorer.handle, float(alpha), float(beta)))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                _native.lib.ctcd_destroy(h)
            except Exception:
                pass
            self._handle = None


class DecodePipeline(object):
    """A serving loop's launches kept ``inflight`` at a time, each on a HIP stream and a decoder of its own (round 5).

    One utterance occupies one workgroup -- one CU in the kernel's default build -- and the time axis of an utterance cannot
    be split, so (a) a batch with fewer utterances than the GPU has CUs leaves the other CUs idle for the whole launch, and
    (b) any launch lasts as long as its slowest utterance (the ones with the most exact ``std::nth_element`` replays: 10-15 %
    above the mean on random rows) while the CUs of the finished ones idle.  With two launches in flight the next batch's
    workgroups take those CUs: the rule is ``inflight_for(batch)`` = 2 below the CU count (``min_shard`` is the same rule
    for ranks: ctcdecode_amd/distributed.py), and 2 is also what hides the stragglers of full batches.  No CU is shared:
    this is the default build of the kernel (``cu_sharing=True`` asks for the two-workgroups-per-CU build instead, which
    pays off from three full batches in flight: bench.py ``pipelined``).

        pipe = DecodePipeline(lambda: CTCBeamDecoder(labels, beam_width=100, log_probs_input=True), inflight=2)
        tickets = [pipe.submit(batch) for batch in batches]      # asynchronous; at most `inflight` are in flight
        results = [pipe.result(t) for t in tickets]              # (output, scores, timesteps, out_lens) in HBM, in order

    ``submit`` waits for the slot's previous launch if its status has not been looked at yet; tickets stay fetchable in any
    order.  A batch that failed (a status word, a scorer callback's exception) raises from ``result`` of ITS ticket -- every time it
    is asked -- and from ``drain``; it never blocks its slot: the next ``submit`` there goes ahead.
    """

    def __init__(self, make_decoder, inflight=2, cu_sharing=False):
        if inflight < 1:
            raise ValueError("inflight must be at least 1")
        self._decs = [make_decoder() for _ in range(inflight)]
        self._device = self._decs[0]._device
        for d in self._decs:
            if d._device != self._device:
                raise ValueError("DecodePipeline: every decoder must sit on the same device")
            if cu_sharing:
                d.set_cu_sharing(1)
        self._streams = [torch.cuda.Stream(device=self._device) for _ in range(inflight)]
        self._pending = [None] * inflight  # per slot: the ticket whose status has not been looked at
        self._serial = 0

    @staticmethod
    def inflight_for(batch, device=None):
        """2 when a batch leaves CUs idle (fewer utterances than CUs), else 1 -- the partition rule of DESIGN.md section 7."""
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        return 2 if batch < torch.cuda.get_device_properties(dev).multi_processor_count else 1

    def submit(self, probs, seq_lens=None):
        slot = self._serial % len(self._decs)
        if self._pending[slot] is not None:
            self._finish(self._pending[slot])
        dec, stream = self._decs[slot], self._streams[slot]
        stream.wait_stream(torch.cuda.current_stream(self._device))  # (the caller's tensors are ready on its own stream)
        with torch.cuda.stream(stream):
            res = dec.decode_device(probs, seq_lens, check=False)
            ev = torch.cuda.Event()
            ev.record(stream)
        # (the launch reads the caller's tensors on the side stream: they stay referenced until it has finished -- a tensor
        #  dropped by the caller right after submit() would otherwise go back to the caching allocator and be handed out again)
        ticket = {"slot": slot, "serial": self._serial, "res": res, "event": ev, "batch": int(probs.shape[0]), "done": False,
                  "inputs": (probs, seq_lens), "error": None}
        self._pending[slot] = ticket
        self._serial += 1
        return ticket

    def _finish(self, ticket):
        """Waits for the ticket's launch and looks at its status words once; a failure is kept on the ticket (raising from
        here left the slot pending for ever)."""
        if ticket["done"]:
            return
        slot = ticket["slot"]
        try:
            ticket["event"].synchronize()
            self._decs[slot]._check(_native.lib.ctcd_check_status(self._decs[slot]._handle, ticket["batch"]))
        except Exception as e:  # noqa: BLE001 (whatever decode_device would have raised)
            ticket["error"] = e
        finally:
            ticket["done"] = True
            ticket["inputs"] = None
            if self._pending[slot] is ticket:
                self._pending[slot] = None

    def result(self, ticket):
        """The four HBM tensors of a submitted batch (waits for its launch; raises what ``decode_device`` would have raised)."""
        self._finish(ticket)
        if ticket.get("error") is not None:
            raise ticket["error"]
        torch.cuda.current_stream(self._device).wait_event(ticket["event"])
        return ticket["res"]

    def drain(self):
        """Waits for everything in flight; raises the first failure among the batches nobody has asked about yet."""
        first = None
        for t in list(self._pending):
            if t is not None:
                self._finish(t)
                if first is None and t.get("error") is not None:
                    first = t["error"]
        if first is not None:
            raise first


class OnlineCTCBeamDecoder(object):
    """Streaming drop-in for ctcdecode/__init__.py:143-250 (no language model): feed an utterance chunk by chunk through a
    ``DecoderState``; results are produced for the items whose ``is_eos_s`` entry is True.  The beam and node pool of
    every stream stay in HBM between calls; ``timesteps`` count frames from the beginning of the stream."""

    def __init__(self, labels, model_path=None, alpha=0, beta=0, cutoff_top_n=40, cutoff_prob=1.0, beam_width=100,
                 num_processes=4, blank_id=0, log_probs_input=False, device=None, logits_input=False, scorer=None):
        self._cutoff_top_n = cutoff_top_n
        self._beam_width = beam_width
        self._scorer = None
        self._num_processes = num_processes
        self._labels = list(labels)
        self._num_labels = len(labels)
        self._blank_id = blank_id
        # logits_input (extension, not in the reference): the input holds raw logits; the device normalises them with a
        # float32 log_softmax (include/ctcdecode_amd.h ctcd_log_softmax) before decoding
        self._log_probs = 2 if logits_input else (1 if log_probs_input else 0)
        self._cutoff_prob = cutoff_prob
        if not torch.cuda.is_available():
            raise RuntimeError("ctcdecode_amd: no HIP device visible; this decoder has no CPU path")
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        h = ctypes.c_void_p()
        _native.check(_native.lib.ctcd_create(ctypes.byref(h), self._device.index))
        self._handle = h
        if scorer is not None:
            self._scorer = _adopt_scorer(scorer, model_path, self._num_labels, self._device.index)
        elif model_path:  # ctcdecode/__init__.py:183-187
            self._scorer = _Scorer(alpha, beta, model_path, self._labels, self._device.index)

    def _check(self, rc):
        """``_native.check`` for the calls a scorer callback runs under: an exception raised inside the callback comes first."""
        if rc and self._scorer is not None and hasattr(self._scorer, "_raise_pending"):
            self._scorer._raise_pending()
        _native.check(rc)

    def set_threads(self, n):
        """Test hook (as CTCBeamDecoder.set_threads): threads per workgroup, 0 = the library's choice."""
        _native.check(_native.lib.ctcd_set_threads(self._handle, int(n)))

    def set_scorer_wait(self, on=True):
        """As CTCBeamDecoder.set_scorer_wait."""
        _native.check(_native.lib.ctcd_set_scorer_wait(self._handle, 1 if on else 0))

    def decode(self, probs, states, is_eos_s, seq_lens=None, check=True):
        """Same contract as ctcdecode/__init__.py:189-238: returns CPU tensors (beam_results[B, R, L], beam_scores[B, K],
        timesteps[B, R, L], out_lens[B, K]) with R = most results of any item that ended (0 if none), L = longest beam.

        ``check=False`` (extension): a call in which no stream ends returns as soon as the chunk's kernel is queued on the
        current stream instead of waiting for its status words -- a serving loop feeds chunk after chunk without a host
        synchronisation in between; a failure surfaces at the next checked call (every call that ends a stream is one)."""
        if probs.dim() != 3:
            raise ValueError("probs must be [batch, time, labels]")
        B, T, V = probs.shape
        if V != self._num_labels:
            raise ValueError("probs.shape[2] (%d) does not match the number of labels (%d)" % (V, self._num_labels))
        if len(states) != B or len(is_eos_s) != B:
            raise ValueError("states and is_eos_s need one entry per batch item")
        probs = probs.to(device=self._device, dtype=torch.float32).contiguous()
        lens_cpu = None
        if seq_lens is not None:
            lens_cpu = seq_lens.detach().cpu().to(torch.int32).contiguous()
        K = self._beam_width
        # (a serving loop calls this once per chunk with the same states: the per-call host work is kept small -- the array of
        #  state handles is rebuilt only when the list changes, nothing is allocated or read back unless a stream ends)
        # (the cache holds WEAK references: a strong list kept ended streams' HBM blocks alive until the next call with other
        #  states, and tied decoder and states into a cycle of objects with finalisers -- ADVICE r4)
        cache = getattr(self, "_ptr_cache", None)
        if cache is None or len(cache[0]) != B or any(r() is not st for r, st in zip(cache[0], states)):
            cache = ([weakref.ref(st) for st in states], (ctypes.c_void_p * max(B, 1))(*[st._ptr(self) for st in states]))
            self._ptr_cache = cache
        ptrs = cache[1]
        any_eos = any(is_eos_s)
        eos = (ctypes.c_ubyte * max(B, 1)).from_buffer_copy(bytes(bytearray(1 if e else 0 for e in is_eos_s)) or b"\0")
        out_T = 0
        if any_eos:
            for b in range(B):
                if is_eos_s[b]:
                    ln = T if lens_cpu is None else max(0, min(int(lens_cpu[b]), T))
                    out_T = max(out_T, int(_native.lib.ctcd_stream_frames(ptrs[b])) + ln)
        if any_eos and not (self._scorer is not None and isinstance(self._scorer, CallbackScorer)) and out_T <= 65536 and V <= 65535:
            # streams end: their results leave the GPU as compact records and are expanded on the host, straight into tensors of the
            # reference's sizes (binding.cpp:186-205) -- the allocator below is called once the kernel has run and R, L are known
            made = {}

            def alloc(_user, R, L, p_tok, p_ts):
                try:
                    pin = B * R * L > 0
                    made["tok"] = torch.empty((B, R, L), dtype=torch.int32, pin_memory=pin)
                    made["ts"] = torch.empty((B, R, L), dtype=torch.int32, pin_memory=pin)
                    p_tok[0] = made["tok"].data_ptr()
                    p_ts[0] = made["ts"].data_ptr()
                    return 0
                except Exception as e:  # (reported through the return code: no exception crosses the C frame)
                    made["error"] = e
                    return 1

            cb = _native.RESULT_ALLOC_FN(alloc)
            scores = torch.empty((B, K), dtype=torch.float32, pin_memory=True)
            out_len = torch.empty((B, K), dtype=torch.int32, pin_memory=True)
            nres = torch.empty((B,), dtype=torch.int32, pin_memory=True)
            R, L = ctypes.c_int(0), ctypes.c_int(0)
            with torch.cuda.device(self._device):
                stream = torch.cuda.current_stream(self._device).cuda_stream
                rc = _native.lib.ctcd_stream_decode_to_host(
                    self._handle, ptrs, eos, probs.data_ptr(), lens_cpu.data_ptr() if lens_cpu is not None else None, B, T, V, K, self._num_processes,
                    float(self._cutoff_prob), int(self._cutoff_top_n), int(self._blank_id), self._log_probs, cb, None,
                    scores.data_ptr(), out_len.data_ptr(), nres.data_ptr(), out_T, ctypes.byref(R), ctypes.byref(L), stream)
            self._ptr_cache = None  # (ended streams are not decoded again: the next call brings other states)
            if "error" in made:
                raise made["error"]
            self._check(rc)
            return made["tok"], scores, made["ts"], out_len
        with torch.cuda.device(self._device):
            scr = getattr(self, "_scratch", None)
            if scr is None or scr[0].shape[0] != B:
                scr = (torch.empty((B, K), dtype=torch.float32, device=self._device), torch.empty((B, K), dtype=torch.int32, device=self._device),
                       torch.empty((B,), dtype=torch.int32, device=self._device))
                self._scratch = scr
            scores, out_len, nres = scr
            output = torch.empty((B, K, out_T), dtype=torch.int32, device=self._device)
            timesteps = torch.empty((B, K, out_T), dtype=torch.int32, device=self._device)
            stream = torch.cuda.current_stream(self._device).cuda_stream
            self._check(_native.lib.ctcd_stream_decode(
                self._handle, ptrs, eos, probs.data_ptr(), lens_cpu.data_ptr() if lens_cpu is not None else None, B, T, V, K, self._num_processes,
                float(self._cutoff_prob), int(self._cutoff_top_n), int(self._blank_id), self._log_probs,
                output.data_ptr(), timesteps.data_ptr(), scores.data_ptr(), out_len.data_ptr(), nres.data_ptr(), out_T, stream))
            if check or any_eos:
                _native.check(_native.lib.ctcd_check_status(self._handle, B))
        if any_eos:
            self._ptr_cache = None  # (ended streams are not decoded again: the next call brings other states)
        if not any_eos:  # nothing ended: no results (binding.cpp:186-205 sizes them to the most results of any item: none)
            return (torch.zeros((B, 0, 0), dtype=torch.int32), torch.zeros((B, K), dtype=torch.float32),
                    torch.zeros((B, 0, 0), dtype=torch.int32), torch.zeros((B, K), dtype=torch.int32))
        nres_c = nres.cpu()
        out_len_c = out_len.cpu()
        R = int(nres_c.max()) if B else 0          # binding.cpp:186-205: sized to the most results / the longest beam
        L = int(out_len_c.max()) if B and R else 0
        return _to_host((output[:, :R, :L].contiguous(), scores, timesteps[:, :R, :L].contiguous(), out_len))

    def character_based(self):
        return bool(_native.lib.ctcd_scorer_is_character_based(self._scorer.handle)) if self._scorer else None

    def max_order(self):
        return int(_native.lib.ctcd_scorer_max_order(self._scorer.handle)) if self._scorer else None

    def dict_size(self):
        return int(_native.lib.ctcd_scorer_dict_size(self._scorer.handle)) if self._scorer else None

    def reset_params(self, alpha, beta):
        if self._scorer is not None:
            _native.check(_native.lib.ctcd_scorer_reset_params(self._scorer.handle, float(alpha), float(beta)))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                _native.lib.ctcd_destroy(h)
            except Exception:
                pass
            self._handle = None


class DecoderState(object):
    """State of one audio stream (ctcdecode/__init__.py:253-272).  Bound to the decoder it is first used with; not reusable
    after its stream ended, as in the reference."""

    def __init__(self, decoder):
        self._decoder = decoder
        h = ctypes.c_void_p()
        sc = getattr(decoder, "_scorer", None)  # ctcdecode/__init__.py:255-269: the state is created with the decoder's scorer
        _native.check(_native.lib.ctcd_stream_create_lm(decoder._handle, ctypes.byref(h), decoder._num_labels, decoder._beam_width, 0,
                                                        sc.handle if sc is not None else None))
        self.state = h

    def _ptr(self, decoder):
        if decoder is not self._decoder:
            raise ValueError("DecoderState used with a different decoder than it was created for")
        return self.state.value

    def __del__(self):
        h = getattr(self, "state", None)
        d = getattr(self, "_decoder", None)
        if h is not None and h.value and d is not None and getattr(d, "_handle", None) is not None:
            try:
                _native.lib.ctcd_stream_destroy(d._handle, h)
            except Exception:
                pass
            self.state = None
