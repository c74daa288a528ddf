"""The one attention backend: fused RoPE + KV-cache write, then the MFMA paged
attention kernels (decode form for the leading decode sequences of the batch,
tiled form for the prefill sequences).

Reference: vllm/attention/layer.py:27-481 (Attention module, KV-connector
hooks :380-395), vllm/v1/attention/backends/triton_attn.py:189-449 (the ROCm
default backend whose role this plays).  There is no backend selector here.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..forward_context import AttentionMetadata, get_forward_context
from ..sched.kv_cache_interface import FullAttentionSpec, KVCacheSpec, MLASpec, SlidingWindowSpec


NATIVE_HEAD_SIZES = (64, 96, 128, 192, 256)


class Attention(nn.Module):
    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 alibi_slopes: Optional[list[float]] = None, sliding_window: Optional[int] = None,
                 logits_soft_cap: Optional[float] = None, prefix: str = "", cache_dtype: str = "auto",
                 causal: bool = True, local_chunk: int = 0, kv_source: Optional["Attention"] = None,
                 blocksparse: Optional[dict] = None):
        super().__init__()
        # Block-sparse causal pattern (Phi-3-small): a query block sees the last ``local_blocks`` key blocks and, per
        # head, every ``vert_stride``-th key block.  Keys: block_size, local_blocks, vert_stride, homo_head,
        # total_heads, first_head (this rank's first head).  Sequences that fit the local span see every causal key
        # and take the MFMA kernels; longer ones run the masked PyTorch form over the paged cache (eager only).
        self.blocksparse = dict(blocksparse) if blocksparse else None
        # A layer that shares another layer's keys and values (Gemma-3n's last layers): it owns no cache and
        # writes none, its queries read the source layer's cache, which that layer filled earlier in the same
        # step (reference: vllm/attention/layer.py kv_sharing_target_layer_name).  Kept in a list so that the
        # source is not registered as a submodule.
        self._kv_source = [kv_source] if kv_source is not None else None
        # > 0: chunked local attention (Llama-4 rotary layers): position p sees positions
        # [p - p % local_chunk, p] only; the runner supplies the per-chunk virtual batch
        # (reference: vllm/v1/attention/backends/utils.py:242 make_local_attention_virtual_batches)
        self.local_chunk = int(local_chunk or 0)
        # The MFMA kernels are built for heads of 64, 96, 128, 192 and 256 values (96 and 192 with 16-bit caches).
        # Other sizes (80, 112, 160, ...) run zero-padded to the next of these: padding adds nothing to q.k, and
        # the padded output values are zero and cut off again (reference head sizes:
        # csrc/attention/paged_attention_v1.cu:150-188).
        self.true_head_size = int(head_size)
        from ..config import get_current_vllm_config_or_none
        cfg = get_current_vllm_config_or_none()
        if cache_dtype == "auto" and cfg is not None and getattr(cfg, "cache_config", None) is not None:
            cache_dtype = cfg.cache_config.cache_dtype        # models that do not pass it on
        sizes = (64, 128) if str(cache_dtype).startswith("fp8") else NATIVE_HEAD_SIZES
        if head_size not in sizes:
            if head_size > sizes[-1]:
                raise ValueError(f"head size {head_size} is beyond the MFMA attention kernels (at most {sizes[-1]}"
                                 f"{' with an fp8 cache' if len(sizes) == 2 else ''})")
            head_size = next(d for d in sizes if d >= head_size)
        self.num_heads, self.head_size, self.scale = num_heads, head_size, float(scale)
        self.causal = causal          # False: encoder models, every token sees its whole sequence
        self.num_kv_heads = num_kv_heads or num_heads
        self.sliding_window = sliding_window or 0
        self.softcap = float(logits_soft_cap or 0.0)
        self.layer_name = prefix
        self.cache_dtype = cache_dtype
        # fp8 KV cache scales: stored value = x / scale (1.0 unless the checkpoint carries calibrated scales)
        self._k_scale_float = self._v_scale_float = 1.0
        qc = getattr(cfg, "quant_config", None) if cfg is not None else None
        self.quant_method = qc.get_quant_method(self, prefix) if qc is not None else None
        if self.quant_method is not None:
            self.quant_method.create_weights(self)
        if alibi_slopes is not None:
            self.register_buffer("alibi_slopes", torch.tensor(alibi_slopes, dtype=torch.float32), persistent=False)
        else:
            self.alibi_slopes = None
        # bound by the model runner after KV memory planning
        self.k_cache: Optional[torch.Tensor] = None
        self.v_cache: Optional[torch.Tensor] = None

    def get_kv_cache_spec(self, block_size: int, dtype: torch.dtype) -> KVCacheSpec:
        if self.sliding_window:
            return SlidingWindowSpec(block_size=block_size, num_kv_heads=self.num_kv_heads,
                                     head_size=self.head_size, dtype=dtype, sliding_window=self.sliding_window)
        return FullAttentionSpec(block_size=block_size, num_kv_heads=self.num_kv_heads, head_size=self.head_size,
                                 dtype=dtype)

    def bind_kv_cache(self, k_cache: torch.Tensor, v_cache: torch.Tensor) -> None:
        self.k_cache, self.v_cache = k_cache, v_cache

    def _forward_blocksparse(self, out, q, md, n, ksc, vsc) -> None:
        from ..ops import reference
        bp = self.blocksparse
        bs, local, stride = int(bp["block_size"]), int(bp["local_blocks"]), int(bp["vert_stride"])
        step = 0 if bp.get("homo_head", False) else max(1, stride // int(bp["total_heads"]))
        heads = torch.arange(self.num_heads, device=q.device) + int(bp.get("first_head", 0))

        def visible(pos, key):
            qb, kb = pos // bs, key // bs
            near = (qb - kb) < local                                             # [ql, sl]
            stripe = ((kb[None] + heads[:, None, None] * step + 1) % stride) == 0      # [H, 1, sl]
            return near[None] | stripe
        if torch.cuda.is_available() and q.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("block-sparse attention beyond the local span cannot be captured in a graph")
        S = md.num_decode_seqs + md.num_prefill_seqs
        out[:n] = reference.paged_attention(q[:n], self.k_cache, self.v_cache, md.block_table[:S], md.seq_lens[:S],
                                            md.query_start_loc[:S + 1], self.scale, k_scale=ksc, v_scale=vsc,
                                            visible=visible)

    @property
    def kv_source(self) -> Optional["Attention"]:
        return self._kv_source[0] if self._kv_source is not None else None

    def _forward_shared(self, qkv, positions, rotary, md, out):
        """Queries only (``qkv`` holds [T, Hq * D]): rotate them and attend over the source layer's cache."""
        src = self._kv_source[0]
        T, n = qkv.shape[0], md.num_actual_tokens
        Hq, D = self.num_heads, self.head_size
        q = qkv[:, : Hq * D]
        if rotary is not None:
            rotary(positions[:T], q, None)
        q = q.reshape(T, Hq, D)
        nd, ndt, n_pre = md.num_decode_seqs, md.num_decode_tokens, md.num_prefill_seqs
        ksc, vsc = src._k_scale_float, src._v_scale_float
        if nd > 0:
            ops.paged_attention_decode(out[:ndt], q[:ndt], src.k_cache, src.v_cache, md.block_table[:nd],
                                       md.seq_lens[:nd], md.query_start_loc[:nd + 1], self.scale,
                                       num_splits=md.decode_num_splits, max_q_len=md.max_decode_query_len,
                                       window=self.sliding_window, softcap=self.softcap, alibi=self.alibi_slopes,
                                       workspace=md.decode_workspace, k_scale=ksc, v_scale=vsc)
        if n_pre > 0:
            e = ndt + md.num_prefill_tokens
            ops.paged_attention_prefill(out[ndt:e], q[ndt:e], src.k_cache, src.v_cache, md.block_table[nd:nd + n_pre],
                                        md.seq_lens[nd:nd + n_pre], md.prefill_query_start_loc, self.scale,
                                        window=self.sliding_window, softcap=self.softcap, alibi=self.alibi_slopes,
                                        k_scale=ksc, v_scale=vsc, causal=self.causal, max_q_len=md.max_query_len)
        if n < T:
            out[n:].zero_()
        return out.view(T, Hq * D)

    def forward(self, qkv: torch.Tensor, positions: Optional[torch.Tensor] = None, rotary=None) -> torch.Tensor:
        """qkv: [T, (Hq + 2 Hkv) * D] fused projection output.  Returns [T, Hq * D]."""
        D0, D = self.true_head_size, self.head_size
        if D0 == D:
            return self._forward_native(qkv, positions, rotary)
        if self._kv_source is not None:
            raise NotImplementedError("layers that share another layer's keys need a head size the kernels have")
        T = qkv.shape[0]
        Hq, Hkv = self.num_heads, self.num_kv_heads
        if rotary is not None and positions is not None:
            rotary(positions[:T], qkv[:, : Hq * D0], qkv[:, Hq * D0: (Hq + Hkv) * D0])
        padded = qkv.new_zeros((T, Hq + 2 * Hkv, D))
        padded[:, :, :D0] = qkv.view(T, Hq + 2 * Hkv, D0)
        out = self._forward_native(padded.view(T, -1), None, None)
        return out.view(T, Hq, D)[:, :, :D0].reshape(T, Hq * D0)

    def _forward_native(self, qkv: torch.Tensor, positions: Optional[torch.Tensor] = None, rotary=None) -> torch.Tensor:
        ctx = get_forward_context()
        md: Optional[AttentionMetadata] = ctx.attn_metadata
        T = qkv.shape[0]
        Hq, Hkv, D = self.num_heads, self.num_kv_heads, self.head_size
        qs, ks = Hq * D, Hkv * D
        out = torch.empty((T, Hq, D), dtype=qkv.dtype, device=qkv.device)
        if self._kv_source is not None:
            if md is None or self._kv_source[0].k_cache is None:
                return out.zero_().view(T, qs)
            return self._forward_shared(qkv, positions, rotary, md, out)
        if md is None or self.k_cache is None:
            # memory-profiling run: no cache bound yet, the projections are what matters
            if rotary is not None and positions is not None:
                rotary(positions, qkv[:, :qs], qkv[:, qs:qs + ks])
            out.zero_()
            return out.view(T, qs)
        n = md.num_actual_tokens
        ksc, vsc = self._k_scale_float, self._v_scale_float
        conn = ctx.kv_connector
        if conn is not None:
            conn.wait_for_layer_load(self.layer_name)
        if rotary is not None:
            table, pos = rotary.cos_sin_cache, positions
            if getattr(rotary, "step_table", None) is not None:
                table, pos = rotary.step_table, rotary.step_index      # multimodal rotary: one row per token
            ops.rope_and_cache(pos[:n], qkv[:n], table, self.k_cache, self.v_cache,
                               md.slot_mapping[:n], rotary.rotary_dim, D, Hq, Hkv, rotary.is_neox_style, ksc, vsc)
        else:
            ops.reshape_and_cache(qkv[:n, qs:qs + ks].view(n, Hkv, D), qkv[:n, qs + ks:].view(n, Hkv, D),
                                  self.k_cache, self.v_cache, md.slot_mapping[:n], ksc, vsc)
        q = qkv[:, :qs].view(T, Hq, D)
        nd, ndt = md.num_decode_seqs, md.num_decode_tokens
        if self.local_chunk and md.local_block_table is not None:
            # same tokens, other sequences: one virtual sequence per (sequence, chunk)
            bt, sl, qsl = md.local_block_table, md.local_seq_lens, md.local_query_start_loc
            pqsl, n_pre = md.local_prefill_query_start_loc, md.local_num_prefill_seqs
        else:
            if self.local_chunk and md.max_seq_len > self.local_chunk:
                raise RuntimeError("chunked local attention needs the runner's per-chunk batch layout")
            bt, sl, qsl = md.block_table, md.seq_lens, md.query_start_loc
            pqsl, n_pre = md.prefill_query_start_loc, md.num_prefill_seqs
        if self.blocksparse is not None and \
                md.max_seq_len > self.blocksparse["block_size"] * self.blocksparse["local_blocks"]:
            self._forward_blocksparse(out, q, md, n, ksc, vsc)
        elif md.cascade_prefix_len and not self.sliding_window and self.causal and not self.local_chunk:
            self._cascade(out, q, md, n, ksc, vsc)
        else:
            if nd > 0:
                ops.paged_attention_decode(out[:ndt], q[:ndt], self.k_cache, self.v_cache, bt[:nd],
                                           sl[:nd], qsl[:nd + 1], self.scale,
                                           num_splits=md.decode_num_splits, max_q_len=md.max_decode_query_len,
                                           window=self.sliding_window, softcap=self.softcap, alibi=self.alibi_slopes,
                                           workspace=md.decode_workspace, k_scale=ksc, v_scale=vsc)
            if n_pre > 0:
                e = ndt + md.num_prefill_tokens
                ops.paged_attention_prefill(out[ndt:e], q[ndt:e], self.k_cache, self.v_cache, bt[nd:nd + n_pre],
                                            sl[nd:nd + n_pre], pqsl, self.scale,
                                            window=self.sliding_window, softcap=self.softcap, alibi=self.alibi_slopes,
                                            k_scale=ksc, v_scale=vsc, causal=self.causal,
                                            max_q_len=md.max_query_len)
        if conn is not None:
            conn.save_kv_layer(self.layer_name, (self.k_cache, self.v_cache), md)
        return out.view(T, qs)


def _cascade(self, out, q, md: AttentionMetadata, n: int, ksc: float, vsc: float) -> None:
    """Cascade attention: every sequence of the step begins with the same ``P`` cached keys.  They are
    read once for all query tokens (one bidirectional pseudo-sequence over the shared blocks) instead
    of once per sequence; each sequence's remaining keys are attended through its block table
    shifted behind the prefix, and the two partial results are merged by their log-sum-exp.
    Reference: vllm/v1/attention/backends/flash_attn.py (cascade_attention, use_cascade_attention)."""
    P = md.cascade_prefix_len
    bs = self.k_cache.shape[2]
    S = md.num_decode_seqs + md.num_prefill_seqs
    kw = dict(softcap=self.softcap, k_scale=ksc, v_scale=vsc, return_lse=True)
    pre = torch.empty_like(out[:n])
    _, pre_lse = ops.paged_attention_prefill(pre, q[:n], self.k_cache, self.v_cache, md.block_table[:1],
                                             md.cascade_seq_len, md.cascade_query_start_loc, self.scale,
                                             causal=False, **kw)
    suf = torch.empty_like(out[:n])
    # ALiBi is relative to each query's own position: in the suffix pass the shift cancels; in the
    # prefix pass every key lies before every query, so the kernel's (key - last key) form would be
    # wrong -- sequences with ALiBi never take this path (see the runner's test)
    _, suf_lse = ops.paged_attention_prefill(suf, q[:n], self.k_cache, self.v_cache, md.block_table[:S, P // bs:],
                                             md.cascade_suffix_seq_lens, md.query_start_loc[:S + 1], self.scale,
                                             causal=True, **kw)
    out[:n] = ops.merge_attn_states(pre, pre_lse, suf, suf_lse)


Attention._cascade = _cascade


class DualChunkAttention(Attention):
    """Dual-chunk attention (Qwen2.5-1M): positions never exceed what the model was trained on.  Keys are rotated
    with their position inside their chunk of ``chunk_len = chunk_size - local_size`` tokens.  A query sees
      * the keys of its own chunk, causally, rotated with its own in-chunk position (relative distances exact),
      * the keys of the chunk before, all of them, rotated with in-chunk position + chunk_len (capped at chunk_size),
      * all older keys, rotated with one fixed far position,
    under ONE softmax.  Here that is three passes of the paged prefill kernel over per-(sequence, chunk) virtual
    sequences -- the own-chunk pass is exactly chunked local attention, the other two read the owner's block table
    at other offsets without a causal mask -- merged by their log-sum-exp.  Decode steps take the same three
    launches (one query row per sequence), so they are captured in the decode graphs like any other layer.
    With ``original_max_position_embeddings`` the reference multiplies queries and keys by
    0.1 ln(len / original_max) + 1 (at least 1) of their sequence; the runner supplies that factor per token.
    Reference: vllm/attention/backends/dual_chunk_flash_attn.py (_dual_chunk_flash_attn_prefill_func :705-1145 dense
    branch, _dual_chunk_flash_attn_decoding :1283-1377), vllm/model_executor/layers/rotary_embedding.py:1679-1845.

    Sparse prefill (``sparse``: per head of this rank a (vertical lines, slash lines) budget, a length threshold and
    the number of probe rows): a prefill step of ONE request longer than the threshold first scores the last probe
    rows of every query chunk against all keys before them (probe queries rotated like the dense passes, except that
    the older two passes use an unclamped in-chunk position + chunk_len and in-chunk position + chunk_size), sums the
    softmax over rows (vertical lines: key positions) and over diagonals (slash lines: query minus key distance),
    keeps the best of each per head (the first 30 keys and the nearest 100 diagonals always), and runs the three
    passes over only those lines (``ops.paged_attention_prefill_sparse``: 32-key steps per 128-row tile, bitmap built
    in LDS by the attention kernel itself).  Steps of several requests, decode steps and shorter requests stay dense.
    Reference: dual_chunk_flash_attn.py :606-662 (budgets), :763-1045 (estimate and split per pass),
    _vertical_slash_sparse_attention :1407-1490, csrc/attention/vertical_slash_index.cu."""
    dual_chunk = True

    def __init__(self, *args, chunk_size: int, local_size: int, original_max_position_embeddings: int = 0,
                 sparse: Optional[dict] = None, **kw):
        super().__init__(*args, **kw)
        if self.sliding_window or self.alibi_slopes is not None:
            raise ValueError("dual-chunk attention takes neither a sliding window nor ALiBi")
        self.chunk_size, self.chunk_len = int(chunk_size), int(chunk_size) - int(local_size)
        self.local_chunk = self.chunk_len           # the runner cuts the step into (sequence, chunk) pairs
        self.original_max = int(original_max_position_embeddings or 0)
        self.sparse = None
        if sparse and sparse.get("heads"):
            heads = [(int(v) + (100 if int(v) == 30 else 0), int(sl)) for v, sl in sparse["heads"]]
            if len(heads) != self.num_heads:
                raise ValueError("sparse attention config must list every head of the layer")
            self.sparse = dict(heads=heads, threshold=int(sparse.get("threshold", 32768)),
                               last_q=int(sparse.get("last_q", 64)))

    def _sparse_lines(self, raw_last, rows, chunks, positions, rotary, md, S: int, n: int):
        """Vertical / slash lines of every query chunk of the step's single request.  ``rows``: per chunk the
        (first, end) token rows of the step, ``raw_last``: their probe rows before rotation, concatenated.
        Returns int32 (vert [C, Hq, NV], vcount [C, Hq], slash [C, Hq, NS], scount [C, Hq])."""
        Hq, Hkv, D, L = self.num_heads, self.num_kv_heads, self.head_size, self.chunk_len
        G = Hq // Hkv
        dev = raw_last.device
        table, rd, neox = rotary.cos_sin_cache, rotary.rotary_dim, rotary.is_neox_style
        bs = self.k_cache.shape[2]
        nb = (S + bs - 1) // bs
        keys = self.k_cache[md.block_table[0, :nb].long()].permute(1, 0, 2, 3).reshape(Hkv, nb * bs, D)[:, :S]
        if keys.dtype == torch.float8_e4m3fn:
            keys = keys.to(raw_last.dtype) * self._k_scale_float
        lasts = [min(self.sparse["last_q"], e - b) for b, e in rows]
        idx = torch.cat([torch.arange(e - m, e, device=dev) for (b, e), m in zip(rows, lasts)])
        p_in = positions[idx] % L
        probes = []
        for shift in (None, L, self.chunk_size):            # own chunk, chunk before (unclamped), everything older
            q = raw_last.clone()
            ops.rotary_embedding(p_in if shift is None else p_in + shift, q, None, table, rd, D, neox)
            probes.append(q.view(-1, Hkv, G, D))
        vmax, smax = max(v for v, _ in self.sparse["heads"]), max(sl for _, sl in self.sparse["heads"])
        vsz = torch.tensor([v for v, _ in self.sparse["heads"]], device=dev)
        ssz = torch.tensor([sl for _, sl in self.sparse["heads"]], device=dev)
        verts, vcs, slashes, scs = [], [], [], []
        o = 0
        for (b, e), m, c in zip(rows, lasts, chunks):
            end = S - n + e                                   # keys before (and at) the last probe row
            cut = [0, max(0, (c - 1) * L), c * L, end]        # older | chunk before | own chunk
            vert_score = torch.empty((Hq, end), dtype=torch.float32, device=dev)
            slash_score = torch.zeros((Hq, end), dtype=torch.float32, device=dev)
            for kv in range(Hkv):                              # one KV head at a time bounds the score matrix
                sc = torch.cat([
                    torch.matmul(probes[2 - j][o:o + m, kv].permute(1, 0, 2).reshape(G * m, D).float(),
                                 keys[kv, cut[j]:cut[j + 1]].float().t()) for j in range(3)], -1) * self.scale
                sc = sc.view(G, m, end)
                tri = torch.ones((m, m), dtype=torch.bool, device=dev).tril()
                sc[:, :, end - m:] = sc[:, :, end - m:].masked_fill(~tri, float("-inf"))
                pr = torch.softmax(sc, -1)
                vert_score[kv * G:(kv + 1) * G] = pr.sum(1)
                fl = pr.flip(-1)                                # column t: the key at distance t from the last row
                for r in range(m):
                    off = m - 1 - r
                    slash_score[kv * G:(kv + 1) * G, : end - off] += fl[:, r, off:]
            vert_score[:, :30] = float("inf")
            slash_score[:, :100] = float("inf")
            nv, ns = min(vmax, end), min(smax, end)
            v_idx = torch.topk(vert_score, nv, -1).indices.int()
            s_idx = torch.topk(slash_score, ns, -1).indices.int()
            verts.append(torch.nn.functional.pad(v_idx, (0, vmax - nv)))
            slashes.append(torch.nn.functional.pad(s_idx, (0, smax - ns)))
            vcs.append(vsz.clamp(max=nv).int())
            scs.append(ssz.clamp(max=ns).int())
            o += m
        return (torch.stack(verts).contiguous(), torch.stack(vcs).contiguous(), torch.stack(slashes).contiguous(),
                torch.stack(scs).contiguous())

    def _forward_native(self, qkv: torch.Tensor, positions: Optional[torch.Tensor] = None, rotary=None) -> torch.Tensor:
        ctx = get_forward_context()
        md: Optional[AttentionMetadata] = ctx.attn_metadata
        if md is None or self.k_cache is None:
            return super()._forward_native(qkv, positions, rotary)
        if md.local_block_table is None or md.dca_succ_lens is None or rotary is None:
            raise RuntimeError("dual-chunk attention needs the runner's per-chunk batch layout and a rotary table")
        T = qkv.shape[0]
        Hq, Hkv, D = self.num_heads, self.num_kv_heads, self.head_size
        qs, ks = Hq * D, Hkv * D
        n, L = md.num_actual_tokens, self.chunk_len
        ksc, vsc = self._k_scale_float, self._v_scale_float
        conn = ctx.kv_connector
        if conn is not None:
            conn.wait_for_layer_load(self.layer_name)
        if md.dca_token_scale is not None:
            qkv[:n, :qs + ks] *= md.dca_token_scale[:n]
        raw = qkv[:n, :qs]
        q_succ, q_inter = raw.clone(), raw.clone()
        p_in = positions[:n] % L
        table, rd, neox = rotary.cos_sin_cache, rotary.rotary_dim, rotary.is_neox_style
        S = md.max_seq_len
        sparse = (self.sparse is not None and md.num_decode_seqs == 0 and md.seq_lens.shape[0] == 1
                  and S > self.sparse["threshold"])
        if sparse:
            if torch.cuda.is_available() and qkv.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("sparse dual-chunk prefill sizes its lists on the host: not capturable")
            # the step's query chunks of the one request: chunk number and token rows, on the host
            first = S - n
            chunks = list(range(first // L, (S - 1) // L + 1))
            rows = [(max(c * L, first) - first, min((c + 1) * L, S) - first) for c in chunks]
            m = self.sparse["last_q"]
            raw_last = torch.cat([raw[max(b, e - m):e] for b, e in rows]).clone()
        ops.rope_and_cache(p_in, qkv[:n], table, self.k_cache, self.v_cache, md.slot_mapping[:n], rd, D, Hq, Hkv,
                           neox, ksc, vsc)
        out = torch.empty((T, Hq, D), dtype=qkv.dtype, device=qkv.device)
        if n < T:
            out[n:].zero_()
        qsl = md.local_query_start_loc
        kw = dict(softcap=self.softcap, k_scale=ksc, v_scale=vsc, return_lse=True)
        q = qkv[:n, :qs].view(n, Hq, D)
        if md.dca_passes == 1 and not sparse:
            # no sequence of the step has an older chunk: plain chunked local attention, decode kernel included
            nd, ndt, n_pre = md.num_decode_seqs, md.num_decode_tokens, md.local_num_prefill_seqs
            if nd > 0:
                ops.paged_attention_decode(out[:ndt], q[:ndt], self.k_cache, self.v_cache, md.local_block_table[:nd],
                                           md.local_seq_lens[:nd], qsl[:nd + 1], self.scale,
                                           num_splits=md.decode_num_splits, max_q_len=md.max_decode_query_len,
                                           softcap=self.softcap, workspace=md.decode_workspace, k_scale=ksc,
                                           v_scale=vsc)
            if n_pre > 0:
                e = ndt + md.num_prefill_tokens
                ops.paged_attention_prefill(out[ndt:e], q[ndt:e], self.k_cache, self.v_cache,
                                            md.local_block_table[nd:nd + n_pre], md.local_seq_lens[nd:nd + n_pre],
                                            md.local_prefill_query_start_loc, self.scale, softcap=self.softcap,
                                            k_scale=ksc, v_scale=vsc, causal=True)
        elif md.dca_passes != 1:
            ops.rotary_embedding((p_in + L).clamp_(max=self.chunk_size), q_succ, None, table, rd, D, neox)
            ops.rotary_embedding(torch.full_like(p_in, min(2 * L - 1, self.chunk_size)), q_inter, None, table, rd, D,
                                 neox)
        if sparse:
            lines = self._sparse_lines(raw_last, rows, chunks, positions, rotary, md, S, n)
            i32 = dict(dtype=torch.int32, device=qkv.device)
            q_base = torch.tensor([first + b for b, _ in rows], **i32)
            skw = dict(softcap=self.softcap, k_scale=ksc, v_scale=vsc)
            passes = [(q, md.local_block_table, md.local_seq_lens, [c * L for c in chunks], True)]
            if md.dca_passes != 1:
                passes += [(q_succ.view(n, Hq, D), md.dca_succ_block_table, md.dca_succ_lens,
                            [(c - 1) * L for c in chunks], False),
                           (q_inter.view(n, Hq, D), md.dca_inter_block_table, md.dca_inter_lens, [0] * len(chunks), False)]
            parts = [ops.paged_attention_prefill_sparse(
                torch.zeros_like(out[:n]), qq, self.k_cache, self.v_cache, tbl, lens, qsl, self.scale, *lines,
                torch.tensor(base, **i32), q_base, causal=causal, max_keys=S, **skw)
                for qq, tbl, lens, base, causal in passes]
            if len(parts) == 1:
                out[:n] = parts[0][0]
            else:
                o12, l12 = ops.merge_attn_states(parts[0][0], parts[0][1], parts[1][0], parts[1][1], return_lse=True)
                out[:n] = ops.merge_attn_states(o12, l12, parts[2][0], parts[2][1])
        elif md.dca_passes != 1:
            o1, l1 = ops.paged_attention_prefill(torch.zeros_like(out[:n]), q, self.k_cache, self.v_cache,
                                                 md.local_block_table, md.local_seq_lens, qsl, self.scale,
                                                 causal=True, **kw)
            o2, l2 = ops.paged_attention_prefill(torch.zeros_like(o1), q_succ.view(n, Hq, D), self.k_cache,
                                                 self.v_cache, md.dca_succ_block_table, md.dca_succ_lens, qsl,
                                                 self.scale, causal=False, **kw)
            o3, l3 = ops.paged_attention_prefill(torch.zeros_like(o1), q_inter.view(n, Hq, D), self.k_cache,
                                                 self.v_cache, md.dca_inter_block_table, md.dca_inter_lens, qsl,
                                                 self.scale, causal=False, **kw)
            o12, l12 = ops.merge_attn_states(o1, l1, o2, l2, return_lse=True)
            out[:n] = ops.merge_attn_states(o12, l12, o3, l3)
        if conn is not None:
            conn.save_kv_layer(self.layer_name, (self.k_cache, self.v_cache), md)
        return out.view(T, qs)


class MLAAttention(nn.Module):
    """Multi-head latent attention over a latent KV cache (DeepSeek-V2 / V3).

    The cache keeps one row per token: the normalised latent ``c_kv`` followed by the rotated shared key
    ``k_pe`` (plus a transposed copy of the latent for the P.V product).  Queries arrive with the key
    up-projection already absorbed (``q_abs = q_nope . W_UK``), so all heads attend over the same rows, and
    the output stays in latent space for the caller's value up-projection.  One kernel serves decode and
    (chunked) prefill.
    Reference: vllm/v1/attention/backends/mla/common.py (the "absorbed" decode formulation, :200-330),
    csrc/cache_kernels.cu::concat_and_cache_mla.
    """
    is_mla = True

    def __init__(self, num_heads: int, kv_lora_rank: int, rope_dim: int, scale: float, prefix: str = ""):
        super().__init__()
        self.num_heads, self.kv_lora_rank, self.rope_dim = num_heads, kv_lora_rank, rope_dim
        self.dk, self.dv = kv_lora_rank + rope_dim, kv_lora_rank
        self.scale = float(scale)
        self.layer_name = prefix
        from ..config import get_current_vllm_config_or_none
        cfg = get_current_vllm_config_or_none()
        self.max_model_len = cfg.model_config.max_model_len if cfg is not None else 4096
        self.k_cache: Optional[torch.Tensor] = None       # [blocks, BS, dk]
        self.v_cache: Optional[torch.Tensor] = None       # [blocks, dv, BS]
        self._kv_scale_float = 1.0                        # fp8 cache: stored value = x / scale

    def get_kv_cache_spec(self, block_size: int, dtype: torch.dtype) -> KVCacheSpec:
        # fp8: e4m3 bytes of value / _kv_scale_float (normalised latents and rotated keys sit well inside its range)
        return MLASpec(block_size=block_size, num_kv_heads=1, head_size=self.dk, dtype=dtype, use_mla=True,
                       v_size=self.dv)

    def bind_kv_cache(self, k_cache: torch.Tensor, v_cache: torch.Tensor) -> None:
        self.k_cache, self.v_cache = k_cache, v_cache

    def forward(self, q: torch.Tensor, kv_c: torch.Tensor, k_pe: torch.Tensor) -> torch.Tensor:
        """q [T, H, dk] (absorbed nope part, then rotated rope part); kv_c [T, kv_lora_rank] normalised;
        k_pe [T, rope_dim] rotated.  Returns [T, H, kv_lora_rank]."""
        md: Optional[AttentionMetadata] = get_forward_context().attn_metadata
        T = q.shape[0]
        out = torch.empty((T, self.num_heads, self.dv), dtype=q.dtype, device=q.device)
        if md is None or self.k_cache is None:
            return out.zero_()
        n = md.num_actual_tokens
        ops.mla_cache_write(kv_c[:n], k_pe[:n], self.k_cache, self.v_cache, md.slot_mapping[:n], self._kv_scale_float)
        if n < T:
            out[n:].zero_()
        S = md.num_decode_seqs + md.num_prefill_seqs
        # few query tokens per sequence: split the keys further so the launch fills the chip.  The factor
        # depends on the batch size and the model's maximum length only, so eager steps and captured
        # graphs of one batch size compute identical sums.
        nsplit = 1
        if md.max_query_len <= 4:
            nsplit = ops.decode_num_splits(S * max(self.num_heads // 16, 1), 1, self.max_model_len)
        ops.mla_attention(out[:n], q[:n], self.k_cache, self.v_cache, md.block_table[:S], md.seq_lens[:S],
                          md.query_start_loc[:S + 1], self.scale, md.max_query_len, nsplit, self._kv_scale_float)
        return out
