"""Adapter with the method surface of vLLM's v1 KV connector (``KVConnectorBase_V1``) on top of ``SpeckvKVConnector``.

The reference only sketches how vLLM would use the pool (host/python/vllm_speckv_backend.py:104-129: per token,
``prefetch_step`` per layer and ``get_kv_ptr`` per entry).  vLLM itself moves KV through a *connector*: the scheduler asks how
many tokens of a request already live outside its paged cache, the worker loads them into the paged cache before the forward
pass and saves the new ones after it.  This class offers exactly those entry points, with vLLM's names and argument order, so
that a subclass ``class SpeckvConnector(KVConnectorBase_V1, SpeckvVllmConnector)`` is what an installation would register --
vLLM is not part of this image, so the class is duck-typed (it imports nothing from vLLM) and is exercised by a stand-in for
the scheduler / worker objects (tests/test_gpu_round3.py::test_vllm_shaped_connector_round_trip); differences between vLLM
releases (``block_ids`` as a list or a list of lists per cache group, ``Request.request_id`` / ``req_id``) are absorbed here.

Semantics: a request's KV is keyed by its request id (what a preempted-and-resumed request, or a decode instance that receives
a prompt from a prefill instance, presents again), stored page-wise in the pool in the scheme of the connector
(``fp16`` / ``int8_delta_rle`` exact, ``fp8`` / ``int4`` lossy), and survives until ``free_request``.

Roles.  vLLM creates one connector in the scheduler process and one in every worker process; the only thing that travels
between them is the metadata object of a step.  The state here is split the same way: the scheduler role owns the request
ids (it hands out the engine's integer id of a request and carries it in the metadata), the count of tokens it has had
stored, the prefix a request already has in vLLM's own cache (``num_computed_tokens``), and the list of requests to free; the
worker role owns the pool (``SpeckvKVConnector``), the registered paged caches and what a step has handed over so far.  One
object may play both roles (a single-process engine, the test), or two objects may play one each (``lib=None`` for the
scheduler role: it never touches the GPU) -- tests/test_gpu_round3.py runs both arrangements.

paged cache layout (vLLM v1, FlashAttention backend): one tensor per layer, ``[2, num_blocks, block_size, kv_heads, head_dim]``.
"""
from typing import Dict, List, Optional, Sequence, Tuple

from .kv_connector import SpeckvKVConnector


class ReqMeta:
    """What the worker needs for one request of a step: the engine's id of the request, where the tokens
    [first_token, first_token + num_tokens) sit in the paged cache, and which way they move."""
    __slots__ = ("req_id", "engine_id", "slot_mapping", "first_token", "num_tokens", "is_store")

    def __init__(self, req_id: str, engine_id: int, slot_mapping: List[int], first_token: int, num_tokens: int, is_store: bool):
        self.req_id, self.engine_id, self.slot_mapping = req_id, engine_id, slot_mapping
        self.first_token, self.num_tokens, self.is_store = first_token, num_tokens, is_store


class SpeckvConnectorMetadata:
    """vLLM: ``KVConnectorMetadata`` -- built by the scheduler, bound on the worker for the duration of a step."""

    def __init__(self):
        self.requests: List[ReqMeta] = []
        self.free_engine_ids: List[int] = []               # requests the scheduler has finished with since the last step


def _flat_block_ids(block_ids) -> List[int]:
    if block_ids and isinstance(block_ids[0], (list, tuple)):            # one list per KV cache group (newer releases): group 0
        return list(block_ids[0])
    return list(block_ids or [])


def _req_id(obj) -> str:
    return getattr(obj, "request_id", None) or getattr(obj, "req_id")


def slot_mapping_for(block_ids: Sequence[int], block_size: int, num_tokens: int, first_token: int = 0) -> List[int]:
    """slot of token t = block_ids[t // block_size] * block_size + t % block_size (vLLM's own rule), for the tokens
    [first_token, first_token + num_tokens); `block_ids` are the request's blocks from its token 0 on."""
    return [block_ids[t // block_size] * block_size + t % block_size for t in range(first_token, first_token + num_tokens)]


def _cached_requests(scheduler_output):
    """(request id, new block ids, num_computed_tokens, resumed_from_preemption) of the already-known requests of a step.
    vLLM has shipped this both as a list of per-request objects and as ONE object of parallel lists (``req_ids``,
    ``new_block_ids``, ``num_computed_tokens``, ``resumed_from_preemption``).  A request that was preempted and is scheduled
    again arrives HERE, not in ``scheduled_new_reqs``, with ``resumed_from_preemption`` set and ALL of its (new) block ids."""
    c = getattr(scheduler_output, "scheduled_cached_reqs", None)
    if c is None:
        return []
    if hasattr(c, "req_ids"):
        k = len(c.req_ids)
        nb = getattr(c, "new_block_ids", None) or [None] * k
        nc = getattr(c, "num_computed_tokens", None) or [None] * k
        rs = getattr(c, "resumed_from_preemption", None) or [False] * k
        return [(rid, _flat_block_ids(b) if b else [], n, bool(r)) for rid, b, n, r in zip(c.req_ids, nb, nc, rs)]
    return [(_req_id(r), _flat_block_ids(getattr(r, "new_block_ids", None) or []), getattr(r, "num_computed_tokens", None),
             bool(getattr(r, "resumed_from_preemption", False))) for r in c]


class SpeckvVllmConnector:
    def __init__(self, lib, num_layers: int, num_kv_heads: int = 8, head_dim: int = 128, block_size: int = 16,
                 max_tokens: int = 4096, scheme: str = "fp16"):
        # worker role: the pool.  lib=None builds a scheduler-role object (no GPU, no library)
        self.conn = None if lib is None else SpeckvKVConnector(lib, num_layers=num_layers, num_kv_heads=num_kv_heads, head_dim=head_dim,
                                                                max_tokens=max_tokens, scheme=scheme)
        self.L, self.block_size, self.max_tokens = num_layers, block_size, max_tokens
        # ---- scheduler-role state (never read by the worker-role methods)
        self._ids: Dict[str, int] = {}                     # vLLM request id (a string) -> the engine's request id
        self._next_id = 1
        self._stored: Dict[str, int] = {}                  # tokens of each request the workers have been told to store
        self._computed: Dict[str, int] = {}                # the prefix vLLM's own cache already holds (get_num_new_matched_tokens)
        self._to_load: Dict[str, Tuple[List[int], int, int]] = {}
        self._prompt: Dict[str, Tuple[int, List[int]]] = {}   # a prompt still being prefilled in chunks: (length, blocks so far)
        self._pending_free: List[int] = []
        # ---- worker-role state (never read by the scheduler-role methods)
        self._meta: Optional[SpeckvConnectorMetadata] = None
        self._layers: List[str] = []
        self._caches = {}
        self._live: Dict[int, str] = {}                    # engine ids that have an allocation in this worker's pool
        self._stash: Dict[int, Dict[int, tuple]] = {}      # engine id -> layer index -> (k rows, v rows) waiting for wait_for_save
        self._held = []                                    # sources of asynchronous pool writes (until the next step)

    # ------------------------------------------------------------------ scheduler side
    def _engine_id(self, rid: str) -> int:
        if rid not in self._ids:
            self._ids[rid] = self._next_id
            self._next_id += 1
        return self._ids[rid]

    def get_num_new_matched_tokens(self, request, num_computed_tokens: int) -> Tuple[int, bool]:
        """Tokens of `request` beyond `num_computed_tokens` whose KV can be loaded from the pool (whole blocks only; never
        the request's last token: vLLM recomputes at least one).  Second value: loading is synchronous here."""
        rid = _req_id(request)
        have = self._stored.get(rid, 0)
        total = getattr(request, "num_tokens", None) or len(getattr(request, "prompt_token_ids", []) or [])
        usable = min(have, max(total - 1, 0)) // self.block_size * self.block_size
        self._computed[rid] = num_computed_tokens           # the external tokens are [num_computed, num_computed + matched)
        return max(usable - num_computed_tokens, 0), False

    def update_state_after_alloc(self, request, blocks, num_external_tokens: int):
        if num_external_tokens > 0:
            rid = _req_id(request)
            ids = blocks.get_block_ids() if hasattr(blocks, "get_block_ids") else blocks
            first = self._computed.get(rid, getattr(request, "num_computed_tokens", 0) or 0)
            self._to_load[rid] = (_flat_block_ids(ids), first, num_external_tokens)

    def build_connector_meta(self, scheduler_output) -> SpeckvConnectorMetadata:
        """Loads: the tokens [num_computed, num_computed + n) of a request with a pool hit.  Stores: a prompt is stored
        ONCE, in the step whose chunk completes it (`num_computed_tokens + num_scheduled_tokens >= len(prompt)`; with chunked
        prefill the earlier chunks are still in the request's paged blocks then), from the slots of all its tokens."""
        meta = SpeckvConnectorMetadata()
        meta.free_engine_ids, self._pending_free = self._pending_free, []
        scheduled = getattr(scheduler_output, "num_scheduled_tokens", None) or {}
        for new_req in getattr(scheduler_output, "scheduled_new_reqs", []):
            rid = _req_id(new_req)
            blocks = _flat_block_ids(new_req.block_ids)
            if rid in self._to_load:
                blk, first, n = self._to_load.pop(rid)
                meta.requests.append(ReqMeta(rid, self._engine_id(rid), slot_mapping_for(blk or blocks, self.block_size, n, first), first, n, is_store=False))
            elif rid not in self._stored:
                n = len(new_req.prompt_token_ids)
                done = (getattr(new_req, "num_computed_tokens", 0) or 0) + scheduled.get(rid, n)
                if done >= n:
                    self._emit_store(meta, rid, blocks, n)
                else:
                    self._prompt[rid] = (n, blocks)
        for rid, new_blocks, computed, resumed in _cached_requests(scheduler_output):
            if rid in self._to_load:
                # a preempted request scheduled again (vLLM reports it among the cached requests, `resumed_from_preemption`, with
                # all of its new block ids): its pool hit is loaded exactly like a new request's.  The blocks recorded at
                # update_state_after_alloc are the request's whole table; a release that only passes them here is served too.
                blk, first, n = self._to_load.pop(rid)
                use = blk or list(new_blocks)
                if len(use) * self.block_size < first + n:
                    self._pending_free = meta.free_engine_ids + self._pending_free      # this metadata is never delivered: its frees stay pending
                    raise RuntimeError(f"request {rid!r}: a pool hit of tokens [{first}, {first + n}) but only {len(use)} blocks to load it into")
                meta.requests.append(ReqMeta(rid, self._engine_id(rid), slot_mapping_for(use, self.block_size, n, first), first, n, is_store=False))
                continue
            if rid not in self._prompt:
                continue
            n, blocks = self._prompt[rid]
            blocks = list(new_blocks) if resumed else blocks + list(new_blocks)      # (a resumed request brings its whole table again)
            if computed is not None and computed + scheduled.get(rid, 0) >= n:
                del self._prompt[rid]
                self._emit_store(meta, rid, blocks, n)
            else:
                self._prompt[rid] = (n, blocks)
        if self._to_load:
            # a hit that update_state_after_alloc recorded but no entry of this step's output carries: vLLM would treat those
            # tokens as computed although nothing fills their blocks -- never drop that silently
            lost = sorted(self._to_load)
            self._to_load.clear()
            # (the metadata object dies with this exception -- the engine-side frees it had taken over go back to the
            # pending list, so a caller that survives the error does not leak their pool allocations)
            self._pending_free = meta.free_engine_ids + self._pending_free
            raise RuntimeError(f"pool hits recorded for {lost} but the scheduler output schedules none of them (neither as a new nor "
                               "as a cached / resumed request): their blocks would be attended without ever being filled")
        return meta

    def _emit_store(self, meta, rid, blocks, n):
        meta.requests.append(ReqMeta(rid, self._engine_id(rid), slot_mapping_for(blocks, self.block_size, n), 0, n, is_store=True))
        self._stored[rid] = n                               # true once this step's wait_for_save has run, i.e. before any later step

    def request_finished(self, request, block_ids) -> Tuple[bool, Optional[dict]]:
        """The paged blocks may be freed at once: what the pool holds is its own copy."""
        self._computed.pop(_req_id(request), None)
        self._prompt.pop(_req_id(request), None)
        return False, None

    def free_request(self, req_id: str):
        """Scheduler role: forget the request; its pool allocation goes with the next step's metadata (`free_engine_ids`).
        An object that also plays the worker role frees it at once."""
        self._stored.pop(req_id, None)
        self._computed.pop(req_id, None)
        self._prompt.pop(req_id, None)
        if req_id in self._ids:
            eid = self._ids.pop(req_id)
            if self.conn is not None and eid in self._live:
                self._free_engine_ids([eid])
            else:
                self._pending_free.append(eid)

    # ------------------------------------------------------------------ worker side
    def _need_pool(self, what):
        if self.conn is None:
            raise RuntimeError(f"{what}: this connector was built without a library (scheduler role); the worker role owns the pool")

    def register_kv_caches(self, kv_caches: Dict[str, "object"]):
        self._need_pool("register_kv_caches")
        self._layers = list(kv_caches.keys())              # insertion order = layer order (vLLM registers them that way)
        self._caches = dict(kv_caches)
        if len(self._layers) != self.L:
            raise ValueError(f"{len(self._layers)} KV cache layers registered, the connector was built for {self.L}")

    def _free_engine_ids(self, ids):
        for eid in ids:
            if self._live.pop(eid, None) is not None:
                self.conn.free_request(eid)
            self._stash.pop(eid, None)

    def bind_connector_metadata(self, connector_metadata: SpeckvConnectorMetadata):
        self._meta = connector_metadata
        if self.conn is not None and connector_metadata is not None:
            self._free_engine_ids(connector_metadata.free_engine_ids)

    def clear_connector_metadata(self):
        self._meta = None

    def start_load_kv(self, forward_context=None, **kwargs):
        """Pool -> paged cache for every load request of the bound metadata (fetch + decompress of the pages that hold the
        request's tokens [first_token, first_token + num_tokens), then vLLM's own scatter:
        ``cache.reshape(2, blocks * block_size, -1)[:, slots] = rows``)."""
        import torch
        if self._meta is None:
            return
        self._need_pool("start_load_kv")
        for r in self._meta.requests:
            if r.is_store:
                continue
            if r.engine_id not in self._live:
                raise RuntimeError(f"request {r.req_id!r} (engine id {r.engine_id}) is not in this worker's pool: the scheduler "
                                   "matched tokens that no save of this worker produced")
            have = self.conn.length(r.engine_id)
            if r.first_token + r.num_tokens > have:
                raise RuntimeError(f"request {r.req_id!r}: tokens [{r.first_token}, {r.first_token + r.num_tokens}) asked for, {have} stored")
            slots = torch.tensor(r.slot_mapping, dtype=torch.int64, device="cuda")
            for li, name in enumerate(self._layers):
                cache = self._caches[name]
                flat = cache.reshape(2, cache.shape[1] * cache.shape[2], cache.shape[3], cache.shape[4])
                for kind in (0, 1):
                    rows = self.conn.kv_rows(r.engine_id, li, kind, r.first_token, r.first_token + r.num_tokens)   # [tokens][heads][dim] fp16
                    flat[kind].index_copy_(0, slots, rows.to(flat.dtype))

    def wait_for_layer_load(self, layer_name: str):
        return                                                  # the loads were issued on the forward pass's own stream

    def save_kv_layer(self, layer_name: str, kv_layer, attn_metadata=None, **kwargs):
        """Paged cache -> (held until wait_for_save) for the store requests: vLLM's own gather by slot mapping."""
        import torch
        if self._meta is None:
            return
        li = self._layers.index(layer_name)
        flat = kv_layer.reshape(2, kv_layer.shape[1] * kv_layer.shape[2], kv_layer.shape[3], kv_layer.shape[4])
        for r in self._meta.requests:
            if not r.is_store:
                continue
            slots = torch.tensor(r.slot_mapping, dtype=torch.int64, device="cuda")
            self._stash.setdefault(r.engine_id, {})[li] = (r.req_id, flat[0].index_select(0, slots), flat[1].index_select(0, slots))

    def wait_for_save(self):
        """All layers of the step have been handed over: one asynchronous pool write per request (speckv_ext_write_runs)."""
        import torch
        self._need_pool("wait_for_save")
        self._held = []
        for eid, layers in self._stash.items():
            if len(layers) != self.L:
                raise RuntimeError(f"request {layers[next(iter(layers))][0]!r}: {len(layers)} of {self.L} layers were saved")
            k = torch.stack([layers[i][1] for i in range(self.L)]).to(torch.float16)     # [layers][tokens][heads][dim]
            v = torch.stack([layers[i][2] for i in range(self.L)]).to(torch.float16)
            if eid in self._live:                                # the same id stored again (a re-used request id): replace
                self.conn.free_request(eid)
            self.conn.add_request(eid)
            self._live[eid] = layers[0][0]
            self._held += self.conn.write_prefill(eid, k, v)
        self._stash.clear()

    def get_finished(self, finished_req_ids) -> Tuple[Optional[set], Optional[set]]:
        return None, None
