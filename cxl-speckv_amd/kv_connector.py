"""A vLLM-shaped KV connector on top of libcxlspeckv.so.

The reference stops at a sketch of the decode loop (reference host/python/vllm_speckv_backend.py:104-129: per token,
``prefetch_step`` per layer, ``get_kv_ptr`` per entry) around a single-request allocator.  This class is the piece a
paged-attention layer would talk to for a BATCH of requests:

  add_request / free_request   one allocation per request (the shim layout of ``CxlSpeckvKVAllocator.allocate``:
                               ``[layer][kind][pos][head]``), its geometry and its request id bound in the engine
  write_prefill                the prompt's K / V of every layer, compressed into the pool
  append                       one decode step's new K / V rows of every layer for every request of the batch; rows
                               are stored page-wise (one 4 KiB page = 2 positions x 8 kv heads x 128), so an odd last
                               position lives in a small fp16 tail on the compute GPU until its partner arrives
  begin_step                   one batched speculative look-ahead for the whole batch (one device-side flush)
  block_table                  device addresses of the decompressed pages of a request (paged-attention consumers)
  attend                       the attention of one layer for the whole batch straight from the compressed records
                               (FP8 / INT4 pools: ``speckv_ext_attend_*_batch``), the tail position folded in with
                               the returned log-sum-exp

Only plain device pointers cross into the library; torch is used for device buffers and the tail fold.
"""
from typing import Dict, List, Optional, Sequence

from .speckv_ctypes import SpeckvLib

PAGE = 4096
SCHEMES = {"fp16": 0, "int8": 1, "int8_delta_rle": 2, "int4": 3, "fp8": 4, "mxfp4": 5}
FUSED = (3, 4, 5)              # schemes whose records the fused attention reads directly (speckv_ext_attend_{int4,fp8,mx4}_*)


class _Request:
    """handle, length, and the fp16 K / V rows of an odd last position ([layers][heads][dim]).  The tail is kept as a
    reference into the batch tensor it arrived in (tensor pair + row) and only sliced when somebody asks for it: creating
    512 tensor views per decode step of a 256-sequence batch cost 0.3 ms of host time that the lockstep path never used."""
    __slots__ = ("handle", "length", "_tail")

    def __init__(self, handle):
        self.handle = handle
        self.length = 0            # positions stored or held in the tail
        self._tail = None          # (k rows, v rows, row index) or (k, v, None) for tensors of this request alone

    def set_tail(self, k, v, row=None):
        self._tail = (k, v, row)

    def clear_tail(self):
        self._tail = None

    @property
    def tail_k(self):
        t = self._tail
        return None if t is None else (t[0] if t[2] is None else t[0][t[2]])

    @property
    def tail_v(self):
        t = self._tail
        return None if t is None else (t[1] if t[2] is None else t[1][t[2]])


def _device_index(values):
    """int32 index tensor on the device without a blocking copy (torch.tensor(list, device="cuda") waits for the stream)."""
    import torch
    return torch.tensor(values, dtype=torch.int32).pin_memory().to("cuda", non_blocking=True)


class SpeckvKVConnector:
    def __init__(self, lib: SpeckvLib, num_layers: int, num_kv_heads: int = 8, head_dim: int = 128, max_tokens: int = 4096,
                 scheme: str = "fp8"):
        if num_kv_heads * head_dim * 2 != 2048:
            raise ValueError("the page-wise layout needs num_kv_heads * head_dim == 1024 fp16 elements per position "
                             "(8 kv heads x 128: Llama-3 8B / 70B)")
        if max_tokens % 32:
            raise ValueError("max_tokens must be a multiple of 32 (tile size of the fused attention)")
        self.lib = lib
        self.L, self.H, self.D, self.T = num_layers, num_kv_heads, head_dim, max_tokens
        self.scheme = SCHEMES[scheme] if isinstance(scheme, str) else int(scheme)
        self.requests: Dict[int, _Request] = {}
        self.region_pages = max_tokens // 2                   # pages of one (layer, kind) region
        self._side = None
        # the tails of the last append as ONE tensor per kind ([n][layers][heads][dim]) and the request ids they belong to:
        # a batch that decodes in lockstep folds / pairs its tails without touching 256 per-request views
        self._tail_ids = ()
        self._tail_k = self._tail_v = None
        self._fold_key = self._arg_key = None
        self._plan = None                                     # device buffer of the step's attention plan
        self._plan_bound = 0
        self._fold_rows = self._fold_idx = self._fold_k = self._fold_v = None
        self._fold_n = 0
        # host-side bookkeeping of a decode loop is per STEP, not per layer: `_epoch` moves whenever a length or the set of
        # requests changes, and everything derived from (batch, lengths) -- the attention plan, the fold rows, the ctypes
        # arrays of a batched append -- is keyed by (the caller's id sequence, epoch) instead of rebuilding 256-element tuples
        # for every layer (that alone was ~20 us per attend() call)
        self._epoch = 0
        self._batch_cache = {}
        self._step_cols = {}
        self._plan_stream = None
        self._kscale = self._kscale_inv = None               # set_k_channel_scale

    def set_k_channel_scale(self, scale):
        """Per-(layer, kv head, channel) pre-scale of K, folded into the query: K / scale goes into the pool, q * scale meets it, q.k is
        unchanged.  `scale`: [layers][heads][dim] tensor of POWERS OF TWO (exact in fp16 both ways; kv_accuracy.pow2_channel_scales
        calibrates them from a prompt's K: the channel's max|k| over the head's median channel), or None to switch it off.  What it buys:
        an outlier channel of K no longer sets the block scale of the channels that share its quantisation group -- on KV-like data
        INT4_G32 loses 0.33-0.42 of the attention output instead of 0.57-0.71 when the query weighs those channels (MXFP4, whose limit
        there is the element's one mantissa bit, gains little): profiles/r06d_kv_format_accuracy.txt, tests/test_gpu_accuracy.py.
        Set it before the first write; rows read back through kv_rows() are scaled back."""
        import torch
        if any(r.length for r in self.requests.values()):
            raise ValueError("set_k_channel_scale after positions were written: the pool would hold K in two scalings")
        if scale is None:
            self._kscale = self._kscale_inv = None
            return
        scale = torch.as_tensor(scale, dtype=torch.float32, device="cuda")
        if tuple(scale.shape) != (self.L, self.H, self.D):
            raise ValueError(f"k channel scale must be [layers][heads][dim] = {(self.L, self.H, self.D)}")
        if not bool(torch.all(scale == torch.exp2(torch.round(torch.log2(scale))))):
            raise ValueError("k channel scales must be powers of two (anything else rounds K and q a second time)")
        self._kscale = scale.to(torch.float16).contiguous()
        self._kscale_inv = (1.0 / scale).to(torch.float16).contiguous()

    def calibrate_k_channel_scale(self, k):
        """set_k_channel_scale from a sample of K ([layers][tokens][heads][dim], e.g. a first prompt): per (layer, head, channel) the
        channel's max|k| over the head's median channel, rounded to a power of two (kv_accuracy.pow2_channel_scales, here on the device).
        Returns the scales it set."""
        import torch
        k = torch.as_tensor(k, device="cuda")
        if k.dim() != 4 or tuple(k.shape[0:1] + k.shape[2:]) != (self.L, self.H, self.D) or k.shape[1] == 0:
            raise ValueError(f"k sample must be [layers][tokens][heads][dim] = ({self.L}, n, {self.H}, {self.D})")
        amax = k.abs().to(torch.float32).amax(dim=1)                                 # [layers][heads][dim]
        med = torch.quantile(amax, 0.5, dim=-1, keepdim=True).clamp_min(1e-9)        # (the mean of the two middle channels, as numpy's median)
        scale = torch.exp2(torch.round(torch.log2((amax / med).clamp_min(1e-9)))).clamp(2.0 ** -14, 2.0 ** 14)      # (exact in fp16 both ways)
        self.set_k_channel_scale(scale)
        return scale

    # The library takes a hipStream_t and reads NULL as "the engine's own stream".  torch's default stream IS the NULL
    # stream, so work issued from it goes through a side stream that is ordered after it and that it then waits for.
    class _On:
        def __init__(self, conn, stream):
            import torch
            self.cur = stream if stream is not None else torch.cuda.current_stream()
            self.side = None
            if self.cur.cuda_stream == 0:
                if conn._side is None:
                    conn._side = torch.cuda.Stream()
                self.side = conn._side

        def __enter__(self):
            import torch
            if self.side is not None:
                self.side.wait_stream(self.cur)
                return self.side
            # an explicit stream that is not torch's current one: tensors the caller (or this class) produced on the
            # current stream -- q.contiguous(), the gathered page images -- must be complete before the library reads them
            now = torch.cuda.current_stream()
            if now.cuda_stream != self.cur.cuda_stream:
                self.cur.wait_stream(now)
            return self.cur

        def __exit__(self, *exc):
            if self.side is not None:
                self.cur.wait_stream(self.side)
            return False

    # ------------------------------------------------------------------ requests
    def add_request(self, req_id: int) -> int:
        if req_id in self.requests:
            raise KeyError(f"request {req_id} already exists")
        self.lib.set_compression_scheme(self.scheme)
        h = self.lib.alloc(2 * self.T * self.L * self.H * self.D * 2)
        self.lib.set_layout(h, self.T, self.L, self.H, self.D, 2)
        self.lib.bind_request(req_id, h, 0)
        self.requests[req_id] = _Request(h)
        self._epoch += 1
        self._batch_cache.clear()
        self._step_cols.clear()
        return h

    def free_request(self, req_id: int):
        r = self.requests.pop(req_id)
        self._epoch += 1
        self._batch_cache.clear()
        self._step_cols.clear()
        if req_id in self._tail_ids:
            self._tail_ids, self._tail_k, self._tail_v = (), None, None
        self._arg_key = self._fold_key = None                 # a plan names record addresses: plan again
        self._plan_stream = None                              # (and no early plan on a stream the caller may be done with)
        self.lib.free(r.handle)                               # the binding goes with the handle

    def length(self, req_id: int) -> int:
        return self.requests[req_id].length

    def _page(self, layer: int, kind: int, pos: int) -> int:
        return ((layer * 2 + kind) * self.T + pos) // 2

    def _batch(self, req_ids):
        """Per-batch constants (the request objects, their handles as a ctypes array), cached by the id sequence."""
        import ctypes
        key = tuple(req_ids)
        b = self._batch_cache.get(key)
        if b is None:
            if len(self._batch_cache) > 16:
                self._batch_cache.clear()
            reqs = [self.requests[r] for r in req_ids]
            b = self._batch_cache[key] = (key, reqs, (ctypes.c_uint64 * len(reqs))(*[r.handle for r in reqs]))
        return b

    # ------------------------------------------------------------------ writes
    def write_prefill(self, req_id: int, k, v, stream=None):
        """k, v: [layers][tokens][heads][dim] fp16 CUDA tensors of the prompt.  ONE asynchronous launch for the request
        (speckv_ext_write_runs: the 2 * layers K / V regions are page runs of the allocation, read in place from k and v);
        returns the tensors the launch reads -- hold them until the stream has passed it."""
        r = self.requests[req_id]
        n = k.shape[1]
        if r.length:
            raise ValueError("write_prefill on a request that already has positions")
        even = n & ~1
        if self._kscale_inv is not None:
            k = k * self._kscale_inv[:, None]                  # [layers][tokens][heads][dim] / [layers][1][heads][dim]
        k = k.contiguous(); v = v.contiguous()
        if even:
            layer_bytes = n * self.H * self.D * 2
            firsts = [self._page(layer, kind, 0) for kind in (0, 1) for layer in range(self.L)]
            srcs = [t.data_ptr() + layer * layer_bytes for t in (k, v) for layer in range(self.L)]
            with self._On(self, stream) as st:
                self.lib.write_runs(r.handle, firsts, srcs, even // 2, st.cuda_stream)
        if n & 1:
            r.set_tail(k[:, n - 1].contiguous().clone(), v[:, n - 1].contiguous().clone())
        r.length = n
        self._epoch += 1
        return [k, v]

    def append(self, req_ids: Sequence[int], k_new, v_new, stream=None):
        """One decode step: k_new, v_new [batch][layers][heads][dim] fp16.  A position that completes a pair is written
        together with its partner -- 2 * layers pages per request, ONE asynchronous launch for the whole batch
        (speckv_ext_write_strided_batch); an odd one waits in the tail.  A handful of torch kernels per call, whatever
        the batch size: the tails of a step are kept as views of one gathered tensor."""
        import torch
        if self._kscale_inv is not None:
            k_new = k_new * self._kscale_inv[None]              # [batch][layers][heads][dim]
        pair_b, tail_b = [], []
        for b, rid in enumerate(req_ids):
            r = self.requests[rid]
            if r.length >= self.T:
                raise ValueError(f"request {rid} is full")
            (pair_b if r.length % 2 else tail_b).append(b)
        keep = []
        if pair_b:
            reqs = [self.requests[req_ids[b]] for b in pair_b]
            whole = len(pair_b) == len(req_ids)
            idx = None if whole else _device_index(pair_b)
            if tuple(req_ids[b] for b in pair_b) == self._tail_ids:
                kt, vt = self._tail_k, self._tail_v                                                    # [n][layers][heads][dim]
            else:
                kt = torch.stack([r.tail_k for r in reqs]); vt = torch.stack([r.tail_v for r in reqs])
            with self._On(self, stream) as st:
                # the page images are built ON the stream the write runs on (ordered after the producers of k_new / the
                # tails by _On): a compress kernel must never see a half-built image
                with torch.cuda.stream(st):
                    kn, vn = (k_new, v_new) if whole else (k_new.index_select(0, idx), v_new.index_select(0, idx))
                    # page image of the pair for every (layer, kind): [n][layer][kind][2 positions][heads][dim]
                    pair = torch.stack((torch.stack((kt, kn), dim=2), torch.stack((vt, vn), dim=2)), dim=2).contiguous()
                keep.append(pair)
                step_bytes = pair[0].numel() * 2
                if len(reqs) == 1:
                    self.lib.write_strided(reqs[0].handle, self._page(0, 0, reqs[0].length - 1), self.region_pages, 2 * self.L,
                                           pair.data_ptr(), st.cuda_stream)
                else:
                    import ctypes
                    import numpy as np
                    n_ = len(reqs)
                    handles = self._batch(req_ids)[2] if whole else [r.handle for r in reqs]
                    firsts = np.fromiter((r.length - 1 for r in reqs), dtype=np.uint64, count=n_) // 2       # _page(0, 0, pos) = pos // 2
                    srcs = np.arange(n_, dtype=np.uint64) * np.uint64(step_bytes) + np.uint64(pair.data_ptr())
                    self.lib.write_strided_batch(handles, firsts, srcs, self.region_pages, 2 * self.L, st.cuda_stream)
            for r in reqs:
                r.clear_tail()
            self._tail_ids, self._tail_k, self._tail_v = (), None, None
        if tail_b:
            if len(tail_b) == len(req_ids):                                                             # copies: the caller may reuse k_new
                tk, tv = k_new.clone(memory_format=torch.contiguous_format), v_new.clone(memory_format=torch.contiguous_format)
            else:
                idx = _device_index(tail_b)
                tk = k_new.index_select(0, idx); tv = v_new.index_select(0, idx)
            for i, b in enumerate(tail_b):
                r = self.requests[req_ids[b]]
                r.set_tail(tk, tv, i)
            self._tail_ids, self._tail_k, self._tail_v = tuple(req_ids[b] for b in tail_b), tk, tv
        for rid in req_ids:
            self.requests[rid].length += 1
        self._epoch += 1
        # The next step's attention plan, now: the host is ahead of the GPU here (the step's attention launches are still
        # running), so its handle look-ups and the upload of its descriptors cost the next step nothing.  On the stream the
        # step's attention ran on: the upload is ordered behind the launches that still read the current plan.
        st = self._plan_stream
        if st is not None and self._arg_key is not None and self._arg_key[0][0] == tuple(req_ids) and self.scheme in FUSED:
            # best effort: the append itself has taken effect (lengths, tails, pool write) -- a plan that cannot be made now (the
            # stream is capturing, or was destroyed by its owner) must not make the caller retry it; attend() plans again instead
            # (only those two cases are swallowed -- the capture status is asked of the stream the plan would go to, not
            # of torch's current one -- a real failure of the plan, a bad layout or an exhausted pool, is the caller's to see)
            from .speckv_ctypes import SpeckvError
            try:
                if not self.lib.stream_is_capturing(st.cuda_stream):
                    self.plan_step(req_ids, st)
                    key, reqs, _ = self._batch(req_ids)
                    self._prepare_tails(req_ids, key, reqs, st)   # (and which rows carry a tail into the next step: 40 us of host time off its start)
                else:
                    self._arg_key = self._plan_stream = None
            except SpeckvError as e:
                self._arg_key = self._plan_stream = None
                if e.status != -4:                             # SPECKV_ERR_INVAL: the stream cannot take the plan now (capturing / destroyed): attend() plans again
                    raise
        return keep                                            # sources of the asynchronous writes: hold until the stream passed them

    # ------------------------------------------------------------------ reads
    def begin_step(self, req_ids: Sequence[int], depth_k: int = 0, force: bool = False) -> Optional[int]:
        """Speculative look-ahead of the next positions of every (request, layer): one prefetch batch, one flush.
        For pools whose pages are consumed decompressed (fp16 / int8 schemes).  The fused attention of the FP8 / INT4 / MXFP4
        pools reads the records themselves and never looks into the ring of decoded pages, so for those pools the call does
        nothing (it used to decode 4 positions x layers x requests per step that nobody read: 0.09 ms of a 1 ms decode step,
        profiles/r06_connector_step.txt) unless `force` says that block_table() / access() consumers exist beside it."""
        import numpy as np
        if self.scheme in FUSED and not force:
            return None
        key, robjs, _ = self._batch(req_ids)
        cols = self._step_cols.get((key, depth_k))            # request / layer / depth columns: constants of the batch
        if cols is None:
            if len(self._step_cols) > 16:
                self._step_cols.clear()
            n = len(req_ids) * self.L
            cols = self._step_cols[(key, depth_k)] = (np.repeat(np.asarray(req_ids, dtype=np.uint32), self.L),
                                                      np.tile(np.arange(self.L, dtype=np.uint16), len(req_ids)),
                                                      np.full(n, depth_k, np.uint32))
        reqs, layers, depth = cols
        pos = np.repeat(np.fromiter((max(r.length - 1, 0) for r in robjs), dtype=np.uint32, count=len(robjs)), self.L)
        self.lib.prefetch_batch(reqs, layers, pos, depth)
        return self.lib.prefetch_flush(want_count=False)

    def block_table(self, req_id: int, layer: int, kind: int, pos_begin: int, pos_end: int) -> List[int]:
        """Device addresses of the decompressed rows [pos_begin, pos_end) (stored positions only)."""
        r = self.requests[req_id]
        row = self.H * self.D * 2
        base = ((layer * 2 + kind) * self.T) * row
        return self.lib.access_batch(r.handle, [base + p * row for p in range(pos_begin, min(pos_end, r.length & ~1))])

    def kv_rows(self, req_id: int, layer: int, kind: int, pos_begin: int = 0, pos_end: Optional[int] = None):
        """The stored rows [pos_begin, pos_end) of one layer (default: all of them) as an fp16 tensor [positions][heads][dim]
        (fetch + decompress of the pages that hold them), tail included.  Only those pages are read."""
        import torch
        r = self.requests[req_id]
        pos_end = r.length if pos_end is None else pos_end
        if not 0 <= pos_begin <= pos_end <= r.length:
            raise ValueError(f"rows [{pos_begin}, {pos_end}) of a request with {r.length} positions")
        even = r.length & ~1
        lo, hi = pos_begin & ~1, min((pos_end + 1) & ~1, even)          # whole pages (2 positions each) of the stored part
        out = torch.empty((max(hi, pos_end) - lo, self.H, self.D), dtype=torch.float16, device="cuda")
        if hi > lo:
            with self._On(self, None) as st:
                self.lib.fetch_range(r.handle, self._page(layer, kind, lo), (hi - lo) // 2, out.data_ptr(), False, st.cuda_stream)
        if pos_end > even:                                               # the odd last position lives in the tail
            out[even - lo] = (r.tail_k if kind == 0 else r.tail_v)[layer]
        if kind == 0 and self._kscale is not None:
            out = out * self._kscale[layer][None]                        # back to the caller's scaling (exact: powers of two)
        return out[pos_begin - lo:pos_end - lo]

    PLAN_BUCKET = 512

    def plan_step(self, req_ids: Sequence[int], stream):
        """The attention plan of this decode step for the batch `req_ids` at their current lengths, on `stream` (a torch
        stream, not the default one).  attend() calls it by itself when the batch or the lengths changed; a caller that
        replays captured per-layer calls runs it before every replay.  Returns the length bound the launches are sized for."""
        import ctypes
        import torch
        key, reqs, handles = self._batch(req_ids)
        B = len(reqs)
        lens = [r.length & ~1 for r in reqs]
        bound = min(self.T, max(self.PLAN_BUCKET, (max(lens) + self.PLAN_BUCKET - 1) // self.PLAN_BUCKET * self.PLAN_BUCKET))
        need = self.lib.attend_plan_bytes(B)
        if self._plan is None or self._plan.numel() < need:
            self._plan = torch.zeros(need, dtype=torch.uint8, device="cuda")
        self.lib.attend_batch_plan(handles, (ctypes.c_uint32 * B)(*lens), bound, self._plan.data_ptr(), need, stream.cuda_stream)
        self._arg_key = ((key, self._epoch), stream.cuda_stream)
        self._plan_bound = bound
        return bound

    def _prepare_tails(self, req_ids, key, reqs, st):
        """Which sequences of the batch keep a position outside the pool, and their fp16 rows as one tensor per kind: the same for every
        layer of a step, so it is set up once per (batch, epoch) -- by append(), while the host is ahead of the GPU, for the step that
        follows; by the first attend() otherwise."""
        import torch
        akey = (key, self._epoch)
        if self._fold_key == akey:
            return
        B = len(reqs)
        odd = [b for b, r in enumerate(reqs) if r.length & 1]
        self._fold_key, self._fold_n = akey, len(odd)
        self._fold_rows = self._fold_idx = self._fold_k = self._fold_v = None
        if odd:
            with torch.cuda.stream(st):
                if len(odd) != B:
                    self._fold_rows = _device_index(odd)
                    inv = [-1] * B
                    for i, b in enumerate(odd):
                        inv[b] = i
                    self._fold_idx = _device_index(inv)
                if tuple(req_ids[b] for b in odd) == self._tail_ids:
                    self._fold_k, self._fold_v = self._tail_k, self._tail_v                 # [n][layers][heads][dim]
                else:
                    self._fold_k = torch.stack([reqs[b].tail_k for b in odd]).contiguous()
                    self._fold_v = torch.stack([reqs[b].tail_v for b in odd]).contiguous()

    def attend(self, layer: int, req_ids: Sequence[int], q, sm_scale: float, stream=None):
        """softmax(q.K^T * sm_scale).V of one layer for the batch.  q: [batch][heads][g][dim] fp16 (g query rows per kv
        head, GQA); returns [batch][heads][g][dim] fp32.  Stored positions come straight from the compressed records
        (one launch pair for the batch); the position still waiting for its partner goes along in the same call
        (speckv_ext_attend_planned_tail: folded in by the MXFP4 kernel itself, by one launch inside the call otherwise)."""
        return self.attend_layers(layer, 1, req_ids, q[None], sm_scale, stream)[0]

    def attend_layers(self, layer_begin: int, n_layers: int, req_ids: Sequence[int], q, sm_scale: float, stream=None):
        """The same for n_layers consecutive layers whose query rows exist at once (speckv_ext_attend_planned_layers): q
        [n_layers][batch][heads][g][dim] fp16, returns [n_layers][batch][heads][g][dim] fp32.  One library call; over an MXFP4 pool
        with a batch that fills the chip, one launch."""
        import torch
        if self.scheme not in FUSED:
            raise ValueError("attend() needs an FP8, INT4 or MXFP4 pool; use block_table() / kv_rows() with the other schemes")
        NL, B, H, G, D = q.shape
        if NL != n_layers:
            raise ValueError("q must be [n_layers][batch][heads][g][dim]")
        key, reqs, _ = self._batch(req_ids)
        if self._kscale is not None:
            q = q * self._kscale[layer_begin:layer_begin + n_layers][:, None, :, None, :]
        q = q.contiguous()
        out = torch.empty((NL, B, H, G, D), dtype=torch.float32, device="cuda")
        lse = torch.empty((NL, B, H, G), dtype=torch.float32, device="cuda")
        # One plan per decode step (speckv_ext_attend_batch_plan: handle look-ups and descriptors once, resident on the
        # device), then launch-only calls (speckv_ext_attend_*_planned*).  The length bound moves in steps of
        # PLAN_BUCKET positions, so a caller that captures its per-layer calls into a HIP graph can replay that graph for
        # PLAN_BUCKET decode steps (plan_step() outside the graph, then the replay).
        akey = (key, self._epoch)
        with self._On(self, stream) as st:
            if self._arg_key != (akey, st.cuda_stream):
                self.plan_step(req_ids, st)
            self._plan_stream = st
            self._prepare_tails(req_ids, key, reqs, st)
            n_tail = self._fold_n
            rows = self._fold_rows.data_ptr() if n_tail and self._fold_rows is not None else 0
            idx = self._fold_idx.data_ptr() if n_tail and self._fold_idx is not None else 0
            kt = self._fold_k.data_ptr() if n_tail else 0
            vt = self._fold_v.data_ptr() if n_tail else 0
            if n_layers == 1 and not n_tail:
                self.lib.attend_planned(self.scheme, self._plan.data_ptr(), B, layer_begin, q.data_ptr(), G, self._plan_bound, sm_scale,
                                        out.data_ptr(), lse.data_ptr(), st.cuda_stream)
            elif n_layers == 1:
                self.lib.attend_planned_tail(self.scheme, self._plan.data_ptr(), B, layer_begin, q.data_ptr(), G, self._plan_bound, sm_scale,
                                             out.data_ptr(), lse.data_ptr(), n_tail, rows, idx, kt, vt, self.L * self.H * self.D, st.cuda_stream)
            else:
                self.lib.attend_planned_layers(self.scheme, self._plan.data_ptr(), B, layer_begin, n_layers, q.data_ptr(), G, self._plan_bound, sm_scale,
                                               out.data_ptr(), lse.data_ptr(), st.cuda_stream, n_tail, rows, idx, kt, vt, self.L * self.H * self.D)
        return out
