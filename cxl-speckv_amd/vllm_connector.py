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

paged cache layout (vLLM v1, FlashAttention backend): one tensor per layer, ``[2, num_blocks, block_size, kv_heads, head_dim]``.
"""
from typing import Dict, List, Optional, Sequence, Tuple

from .kv_connector import SpeckvKVConnector


class ReqMeta:
    """What the worker needs for one request of a step: where its tokens sit in the paged cache, and which way they move."""
    __slots__ = ("req_id", "slot_mapping", "num_tokens", "is_store")

    def __init__(self, req_id: str, slot_mapping: List[int], num_tokens: int, is_store: bool):
        self.req_id, self.slot_mapping, self.num_tokens, self.is_store = req_id, slot_mapping, num_tokens, is_store


class SpeckvConnectorMetadata:
    """vLLM: ``KVConnectorMetadata`` -- built by the scheduler, bound on the worker for the duration of a step."""

    def __init__(self):
        self.requests: List[ReqMeta] = []


def _flat_block_ids(block_ids) -> List[int]:
    if block_ids and isinstance(block_ids[0], (list, tuple)):            # one list per KV cache group (newer releases): group 0
        return list(block_ids[0])
    return list(block_ids or [])


def _req_id(obj) -> str:
    return getattr(obj, "request_id", None) or getattr(obj, "req_id")


def slot_mapping_for(block_ids: Sequence[int], block_size: int, num_tokens: int) -> List[int]:
    """slot of token t = block_ids[t // block_size] * block_size + t % block_size (vLLM's own rule)."""
    return [block_ids[t // block_size] * block_size + t % block_size for t in range(num_tokens)]


class SpeckvVllmConnector:
    def __init__(self, lib, num_layers: int, num_kv_heads: int = 8, head_dim: int = 128, block_size: int = 16,
                 max_tokens: int = 4096, scheme: str = "fp16"):
        self.conn = SpeckvKVConnector(lib, num_layers=num_layers, num_kv_heads=num_kv_heads, head_dim=head_dim,
                                      max_tokens=max_tokens, scheme=scheme)
        self.L, self.block_size = num_layers, block_size
        self._ids: Dict[str, int] = {}                     # vLLM request id (a string) -> the engine's request id
        self._stored: Dict[str, int] = {}                  # tokens of each request that live in the pool
        self._to_load: Dict[str, Tuple[List[int], int]] = {}
        self._meta: Optional[SpeckvConnectorMetadata] = None
        self._layers: List[str] = []
        self._caches = {}
        self._stash: Dict[str, Dict[int, tuple]] = {}      # req id -> layer index -> (k rows, v rows) waiting for wait_for_save
        self._held = []                                    # sources of asynchronous pool writes (until the next step)

    # ------------------------------------------------------------------ scheduler side
    def get_num_new_matched_tokens(self, request, num_computed_tokens: int) -> Tuple[int, bool]:
        """Tokens of `request` beyond `num_computed_tokens` whose KV can be loaded from the pool (whole blocks only; never
        the request's last token: vLLM recomputes at least one).  Second value: loading is synchronous here."""
        have = self._stored.get(_req_id(request), 0)
        total = getattr(request, "num_tokens", None) or len(getattr(request, "prompt_token_ids", []) or [])
        usable = min(have, max(total - 1, 0)) // self.block_size * self.block_size
        return max(usable - num_computed_tokens, 0), False

    def update_state_after_alloc(self, request, blocks, num_external_tokens: int):
        if num_external_tokens > 0:
            ids = blocks.get_block_ids() if hasattr(blocks, "get_block_ids") else blocks
            self._to_load[_req_id(request)] = (_flat_block_ids(ids), num_external_tokens)

    def build_connector_meta(self, scheduler_output) -> SpeckvConnectorMetadata:
        meta = SpeckvConnectorMetadata()
        for new_req in getattr(scheduler_output, "scheduled_new_reqs", []):
            rid = _req_id(new_req)
            blocks = _flat_block_ids(new_req.block_ids)
            if rid in self._to_load:
                blk, n = self._to_load.pop(rid)
                meta.requests.append(ReqMeta(rid, slot_mapping_for(blk or blocks, self.block_size, n), n, is_store=False))
            elif rid not in self._stored:
                n = len(new_req.prompt_token_ids)
                meta.requests.append(ReqMeta(rid, slot_mapping_for(blocks, self.block_size, n), n, is_store=True))
        self._to_load.clear()
        return meta

    def request_finished(self, request, block_ids) -> Tuple[bool, Optional[dict]]:
        """The paged blocks may be freed at once: what the pool holds is its own copy."""
        return False, None

    # ------------------------------------------------------------------ worker side
    def register_kv_caches(self, kv_caches: Dict[str, "object"]):
        self._layers = list(kv_caches.keys())              # insertion order = layer order (vLLM registers them that way)
        self._caches = dict(kv_caches)
        if len(self._layers) != self.L:
            raise ValueError(f"{len(self._layers)} KV cache layers registered, the connector was built for {self.L}")

    def bind_connector_metadata(self, connector_metadata: SpeckvConnectorMetadata):
        self._meta = connector_metadata

    def clear_connector_metadata(self):
        self._meta = None

    def _engine_id(self, rid: str) -> int:
        if rid not in self._ids:
            self._ids[rid] = len(self._ids) + 1
            self.conn.add_request(self._ids[rid])
        return self._ids[rid]

    def start_load_kv(self, forward_context=None, **kwargs):
        """Pool -> paged cache for every load request of the bound metadata (fetch + decompress of the request's pages, then
        vLLM's own scatter: ``cache.reshape(2, blocks * block_size, -1)[:, slots] = rows``)."""
        import torch
        if self._meta is None:
            return
        for r in self._meta.requests:
            if r.is_store:
                continue
            eid = self._ids[r.req_id]
            slots = torch.tensor(r.slot_mapping, dtype=torch.int64, device="cuda")
            for li, name in enumerate(self._layers):
                cache = self._caches[name]
                flat = cache.reshape(2, cache.shape[1] * cache.shape[2], cache.shape[3], cache.shape[4])
                for kind in (0, 1):
                    rows = self.conn.kv_rows(eid, li, kind)[:r.num_tokens]          # [tokens][heads][dim] fp16
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
            self._stash.setdefault(r.req_id, {})[li] = (flat[0].index_select(0, slots), flat[1].index_select(0, slots))

    def wait_for_save(self):
        """All layers of the step have been handed over: one asynchronous pool write per request (speckv_ext_write_runs)."""
        import torch
        self._held = []
        for rid, layers in self._stash.items():
            if len(layers) != self.L:
                raise RuntimeError(f"request {rid}: {len(layers)} of {self.L} layers were saved")
            k = torch.stack([layers[i][0] for i in range(self.L)]).to(torch.float16)     # [layers][tokens][heads][dim]
            v = torch.stack([layers[i][1] for i in range(self.L)]).to(torch.float16)
            self._held += self.conn.write_prefill(self._engine_id(rid), k, v)
            self._stored[rid] = k.shape[1]
        self._stash.clear()

    def get_finished(self, finished_req_ids) -> Tuple[Optional[set], Optional[set]]:
        return None, None

    # ------------------------------------------------------------------ life cycle
    def free_request(self, req_id: str):
        if req_id in self._ids:
            self.conn.free_request(self._ids.pop(req_id))
        self._stored.pop(req_id, None)
