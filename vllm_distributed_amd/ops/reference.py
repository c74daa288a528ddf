"""Plain PyTorch implementations of every op in this package.

They serve two purposes: (1) the numerics oracle the HIP kernels are tested
against (fp32 math, same layouts), and (2) the CPU plumbing path used when no
GPU is present (role of the reference's csrc/cpu backend for the OPT-125m CPU
configuration).  They are never used for CUDA tensors unless
``VDA_FORCE_TORCH_OPS=1`` is set for debugging.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

FP8_MAX = 448.0


# ---- norms -----------------------------------------------------------------
def rms_norm(x, weight, eps, residual=None, weight_bias: float = 0.0):
    dt = x.dtype
    if residual is not None:
        x = (x.float() + residual.float()).to(dt)
        residual.copy_(x)
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    y = (xf * torch.rsqrt(var + eps)).to(dt)
    return (y.float() * (weight.float() + weight_bias)).to(dt)


def layer_norm(x, weight, bias, eps):
    return F.layer_norm(x.float(), (x.shape[-1],), weight.float(),
                        None if bias is None else bias.float(), eps).to(x.dtype)


# ---- activations -------------------------------------------------------------
def _act(kind: str, x: torch.Tensor, threshold: float = 0.0) -> torch.Tensor:
    if kind == "silu":
        return F.silu(x)
    if kind == "gelu":
        return F.gelu(x)
    if kind in ("gelu_tanh", "gelu_pytorch_tanh", "gelu_new"):
        return F.gelu(x, approximate="tanh")
    if kind == "relu":
        return F.relu(x)
    if kind == "gelu_fast":
        return 0.5 * x * (1.0 + torch.tanh(x * 0.79788456 * (1.0 + 0.044715 * x * x)))
    if kind in ("quick_gelu", "gelu_quick"):
        return x * torch.sigmoid(1.702 * x)
    if kind == "fatrelu":
        return torch.where(x > threshold, x, torch.zeros_like(x))
    if kind == "relu2":
        return F.relu(x).square()
    raise ValueError(f"unknown activation {kind!r}")


def act_and_mul(x, kind="silu", act_first=True, threshold=0.0):
    d = x.shape[-1] // 2
    a, b = x[..., :d].float(), x[..., d:].float()
    y = _act(kind, a, threshold) * b if act_first else a * _act(kind, b, threshold)
    return y.to(x.dtype)


def activation(x, kind, threshold=0.0):
    return _act(kind, x.float(), threshold).to(x.dtype)


# ---- rotary ------------------------------------------------------------------
def _rotate(x, cos, sin, rot_dim, is_neox):
    # x: [T, H, D]; cos/sin: [T, rot_dim/2]
    xr = x[..., :rot_dim].float()
    cos, sin = cos[:, None, :], sin[:, None, :]
    if is_neox:
        x1, x2 = xr[..., : rot_dim // 2], xr[..., rot_dim // 2:]
        out = torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1)
    else:
        x1, x2 = xr[..., 0::2], xr[..., 1::2]
        out = torch.stack((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1).flatten(-2)
    res = x.clone()
    res[..., :rot_dim] = out.to(x.dtype)
    return res


def rotary_embedding(positions, q, k, cos_sin_cache, rot_dim, head_size, is_neox=True):
    """In place on q [T, Hq*D] / k [T, Hkv*D] (k may be None)."""
    cs = cos_sin_cache[positions.long()]
    cos, sin = cs[:, : rot_dim // 2].float(), cs[:, rot_dim // 2:].float()
    T = q.shape[0]
    qv = q.view(T, -1, head_size)
    qv.copy_(_rotate(qv, cos, sin, rot_dim, is_neox))
    if k is not None:
        kv = k.view(T, -1, head_size)
        kv.copy_(_rotate(kv, cos, sin, rot_dim, is_neox))


# ---- cache ---------------------------------------------------------------------
def reshape_and_cache(k, v, k_cache, v_cache, slot_mapping, k_scale: float = 1.0, v_scale: float = 1.0):
    """k, v: [T, Hkv, D]; k_cache [nblk, Hkv, BS, D]; v_cache [nblk, Hkv, D, BS]; fp8 caches hold x / scale."""
    bs = k_cache.shape[2]
    slots = slot_mapping.long()
    keep = slots >= 0
    slots = slots[keep]
    blk, off = slots // bs, slots % bs
    kk, vv = k[keep], v[keep]
    if k_cache.dtype == torch.float8_e4m3fn:
        kk = (kk.float() / k_scale).clamp(-448, 448)
        vv = (vv.float() / v_scale).clamp(-448, 448)
    k_cache[blk, :, off, :] = kk.to(k_cache.dtype)
    v_cache[blk, :, :, off] = vv.to(v_cache.dtype)


def gather_cache(k_cache, v_cache, slots):
    bs = k_cache.shape[2]
    slots = slots.long()
    blk, off = slots // bs, slots % bs
    return k_cache[blk, :, off, :], v_cache[blk, :, :, off]


def copy_blocks(cache, pairs):
    for s, d in pairs.tolist():
        cache[d].copy_(cache[s])


def transfer_blocks(dst, src, dst_ids, src_ids):
    dst[dst_ids.long().to(dst.device)] = src[src_ids.long().to(src.device)].to(dst.device)


# ---- attention -------------------------------------------------------------------
def paged_attention(q, k_cache, v_cache, block_table, seq_lens, q_start_loc, scale,
                    window: int = 0, softcap: float = 0.0, alibi=None, return_lse: bool = False,
                    k_scale: float = 1.0, v_scale: float = 1.0, causal: bool = True, visible=None):
    """Attention of the new tokens of every sequence against its paged K/V (which already contain
    the new tokens); causal, or over every key of the sequence.  q: [T, Hq, D] -> [T, Hq, D].
    ``visible(pos [ql, 1], key [1, sl]) -> bool [Hq, ql, sl]``: an extra per-head visibility pattern."""
    T, Hq, D = q.shape
    Hkv, bs = k_cache.shape[1], k_cache.shape[2]
    G = Hq // Hkv
    out = torch.zeros_like(q)
    lse = torch.full((Hq, T), float("-inf"), dtype=torch.float32, device=q.device)
    qsl = q_start_loc.tolist()
    for s, sl in enumerate(seq_lens.tolist()):
        a, b = qsl[s], qsl[s + 1]
        ql = b - a
        if ql == 0 or sl == 0:
            continue
        nb = (sl + bs - 1) // bs
        blocks = block_table[s, :nb].long()
        k = k_cache[blocks].permute(1, 0, 2, 3).reshape(Hkv, nb * bs, D)[:, :sl].float()
        v = v_cache[blocks].permute(1, 0, 3, 2).reshape(Hkv, nb * bs, D)[:, :sl].float()
        if k_cache.dtype == torch.float8_e4m3fn:
            k, v = k * k_scale, v * v_scale
        k = k.repeat_interleave(G, dim=0)
        v = v.repeat_interleave(G, dim=0)
        qq = q[a:b].permute(1, 0, 2).float()                  # [Hq, ql, D]
        sc = torch.matmul(qq, k.transpose(1, 2)) * scale      # [Hq, ql, sl]
        if softcap > 0:
            sc = softcap * torch.tanh(sc / softcap)
        pos = torch.arange(sl - ql, sl, device=q.device)[:, None]
        key = torch.arange(sl, device=q.device)[None, :]
        if alibi is not None:
            sc = sc + alibi.float()[:, None, None] * (key - pos)[None].float()
        mask = (key <= pos) if causal else torch.ones((ql, sl), dtype=torch.bool, device=q.device)
        if window > 0:
            mask &= key > pos - window
        sc = sc.masked_fill(~mask[None], float("-inf"))
        if visible is not None:
            sc = sc.masked_fill(~visible(pos, key), float("-inf"))
        lse[:, a:b] = torch.logsumexp(sc, dim=-1)
        p = torch.softmax(sc, dim=-1)
        out[a:b] = torch.matmul(p, v).permute(1, 0, 2).to(q.dtype)
    return (out, lse) if return_lse else out


def sparse_step_selection(vert, vcount, slash, scount, key_base: int, q_base: int, ql: int, sl: int, causal: bool):
    """Which 32-key steps each 128-row query tile of one sequence visits, per head: bool [Hq, tiles, steps].
    A step is visited when it holds a selected key position, or when a selected diagonal (query position minus key
    position) crosses it inside the tile's rows; causal passes always visit the tile's own diagonal."""
    Hq = vert.shape[0]
    tiles, steps = (ql + 127) // 128, (sl + 31) // 32
    sel = torch.zeros((Hq, tiles, steps), dtype=torch.bool, device=vert.device)
    t = torch.arange(tiles, device=vert.device)
    qa0 = q_base + t * 128 - key_base                                        # [tiles] first / last row, on the key axis
    qa1 = q_base + torch.clamp((t + 1) * 128, max=ql) - 1 - key_base
    for h in range(Hq):
        k = vert[h, : int(vcount[h])].long() - key_base
        k = k[(k >= 0) & (k < sl)]
        sel[h, :, k // 32] = True
        d = slash[h, : int(scount[h])].long()
        d = d[d >= 0]
        bands = [(qa0[:, None] - d[None], qa1[:, None] - d[None])]
        if causal:
            bands.append((qa0[:, None], qa1[:, None]))
        for a, b in bands:
            a, b = a.clamp(min=0), b.clamp(max=sl - 1)
            for j in range(6):                                              # a band of <= 128 keys touches <= 5 steps
                st = a // 32 + j
                ok = (a <= b) & (st <= b // 32)
                ti = t[:, None].expand_as(st)[ok]
                sel[h, ti, st[ok]] = True
    return sel


def paged_attention_sparse(q, k_cache, v_cache, block_table, seq_lens, q_start_loc, scale, vert, vcount, slash, scount,
                           key_base, q_base, causal: bool = True, softcap: float = 0.0, k_scale: float = 1.0,
                           v_scale: float = 1.0):
    """Vertical-slash sparse attention at 32-key granularity (see ``sparse_step_selection``); vert / slash
    [S, Hq, N] with counts [S, Hq], key_base / q_base [S].  Returns (out [T, Hq, D], lse [Hq, T])."""
    out = torch.zeros_like(q)
    lse = torch.full((q.shape[1], q.shape[0]), float("-inf"), dtype=torch.float32, device=q.device)
    qsl, sls = q_start_loc.tolist(), seq_lens.tolist()
    for s, sl in enumerate(sls):
        a, b = qsl[s], qsl[s + 1]
        if a == b or sl == 0:
            continue
        sel = sparse_step_selection(vert[s], vcount[s], slash[s], scount[s], int(key_base[s]), int(q_base[s]), b - a, sl,
                                    causal)

        def visible(pos, key, sel=sel, lo=sl - (b - a)):
            return sel[:, (pos[:, 0] - lo) // 128][:, :, key[0] // 32]
        o, l = paged_attention(q[a:b], k_cache, v_cache, block_table[s:s + 1], seq_lens[s:s + 1],
                               torch.tensor([0, b - a], dtype=torch.int32, device=q.device), scale, softcap=softcap,
                               return_lse=True, k_scale=k_scale, v_scale=v_scale, causal=causal, visible=visible)
        out[a:b], lse[:, a:b] = torch.nan_to_num(o), l
    return out, lse


def merge_attn_states(prefix, prefix_lse, suffix, suffix_lse):
    """prefix/suffix: [T, H, D]; lse: [H, T]."""
    a = torch.where(torch.isposinf(prefix_lse), torch.full_like(prefix_lse, float("-inf")), prefix_lse)
    b = torch.where(torch.isposinf(suffix_lse), torch.full_like(suffix_lse, float("-inf")), suffix_lse)
    mx = torch.maximum(a, b)
    mu = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    ea, eb = torch.exp(a - mu), torch.exp(b - mu)
    den = ea + eb
    wa = torch.where(den > 0, ea / den, torch.zeros_like(den)).T[..., None]
    wb = torch.where(den > 0, eb / den, torch.zeros_like(den)).T[..., None]
    out = (prefix.float() * wa + suffix.float() * wb).to(prefix.dtype)
    return out, torch.where(den > 0, mu + torch.log(den), torch.full_like(den, float("-inf")))


# ---- sampling ------------------------------------------------------------------
def apply_penalties(logits, prompt_mask, out_counts, rep, freq, pres):
    seen = prompt_mask.bool() | (out_counts > 0)
    r = rep[:, None]
    pen = torch.where(logits > 0, logits / r, logits * r)
    logits.copy_(torch.where(seen, pen, logits))
    logits.sub_(freq[:, None] * out_counts.float())
    logits.sub_(pres[:, None] * (out_counts > 0).float())
    return logits


def topk_topp_mask(logits, top_k=None, top_p=None, min_p=None):
    """In place -inf masking, sort based (the oracle for the radix kernel)."""
    S, V = logits.shape
    if top_k is not None:
        k = top_k.long().clone()
        k[(k <= 0) | (k > V)] = V
        srt = logits.sort(dim=-1, descending=True).values
        thr = srt.gather(1, (k - 1)[:, None])
        logits.masked_fill_(logits < thr, float("-inf"))
    if top_p is not None:
        probs = torch.softmax(logits, dim=-1)
        sp, si = probs.sort(dim=-1, descending=True)
        cum = sp.cumsum(-1)
        # smallest set whose mass reaches p; the first token is always kept
        drop = (cum - sp) >= top_p[:, None]
        drop[:, 0] = False
        thr = torch.where(drop, torch.full_like(sp, 2.0), sp).min(dim=-1, keepdim=True).values
        logits.masked_fill_(probs < thr, float("-inf"))
    if min_p is not None:
        probs = torch.softmax(logits, dim=-1)
        mx = probs.max(dim=-1, keepdim=True).values
        logits.masked_fill_(probs < min_p[:, None] * mx, float("-inf"))
    return logits


def sample(logits, temperature=None, top_k=None, top_p=None, min_p=None, seeds=None, generator=None):
    S, V = logits.shape
    out = logits.argmax(dim=-1)
    if temperature is None:
        return out
    rnd = temperature >= 1e-5
    if not bool(rnd.any()):
        return out
    z = logits.float() / torch.where(rnd, temperature, torch.ones_like(temperature))[:, None]
    z = topk_topp_mask(z, top_k, top_p, min_p)
    probs = torch.softmax(z, dim=-1)
    if seeds is not None:
        draws = torch.empty(S, dtype=torch.long, device=logits.device)
        for i in range(S):
            g = torch.Generator(device=logits.device)
            g.manual_seed(int(seeds[i]) & 0x7FFFFFFFFFFFFFFF)
            draws[i] = torch.multinomial(probs[i], 1, generator=g)[0]
    else:
        draws = torch.multinomial(probs, 1, generator=generator)[:, 0]
    return torch.where(rnd, draws, out)


# ---- fp8 -------------------------------------------------------------------------
def scaled_fp8_quant(x, scale=None):
    if scale is None:
        scale = (x.abs().max().float() / FP8_MAX).clamp(min=1e-12).reshape(1)
    q = (x.float() / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q, scale


def per_token_fp8_quant(x, scale_ub=None):
    amax = x.abs().amax(dim=-1, keepdim=True).float()
    if scale_ub is not None:
        amax = amax.clamp(max=float(scale_ub))
    scale = (amax / FP8_MAX).clamp(min=1.0 / (FP8_MAX * 512.0))
    q = (x.float() / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q, scale


def fp8_dequant(q, scale, dtype):
    return (q.float() * scale).to(dtype)


def per_group_fp8_quant(x, group: int = 128):
    """One dynamic scale per ``group`` consecutive elements of a row (the last group may be short)."""
    K = x.shape[-1]
    ng = -(-K // group)
    xf = x.float()
    pad = ng * group - K
    if pad:
        xf = F.pad(xf, (0, pad))
    g = xf.reshape(*x.shape[:-1], ng, group)
    scale = (g.abs().amax(dim=-1, keepdim=True) / FP8_MAX).clamp(min=1.0 / (FP8_MAX * 512.0))
    q = (g / scale).clamp(-FP8_MAX, FP8_MAX).reshape(*x.shape[:-1], ng * group)[..., :K].to(torch.float8_e4m3fn)
    return q, scale.squeeze(-1)


def expand_block_scales(scale: torch.Tensor, rows: int, cols: int, block: int = 128) -> torch.Tensor:
    """[.., ceil(rows/block), ceil(cols/block)] block scales -> one scale per element [.., rows, cols]."""
    s = scale.repeat_interleave(block, dim=-2).repeat_interleave(block, dim=-1)
    return s[..., :rows, :cols]


def fp8_gemm(xq, w, x_scale, w_scale, out_dtype, bias=None, block: bool = False):
    M, K = xq.shape
    if block:
        xf = xq.float() * x_scale.reshape(M, -1).float().repeat_interleave(128, dim=1)[:, :K]
        wf = w.float() * expand_block_scales(w_scale.float(), w.shape[0], K)
    else:
        xf = xq.float() * (x_scale.reshape(-1, 1).float() if x_scale.numel() == M else x_scale.reshape(()).float())
        wf = w.float() * w_scale.reshape(()).float()
    y = xf @ wf.t()
    if bias is not None:
        y = y + bias.float()
    return y.to(out_dtype)


def skinny_gemm(x, w, bias=None, w_scale=None):
    if w.dtype == torch.float8_e4m3fn:
        w = (w.float() * (w_scale.float() if w_scale is not None else 1.0)).to(x.dtype)
    return F.linear(x, w, bias)


# ---- mixture of experts ------------------------------------------------------------------
def moe_topk_softmax(logits: torch.Tensor, k: int, renormalize: bool):
    probs = torch.softmax(logits.float(), dim=-1)
    w, ids = torch.topk(probs, k, dim=-1)
    if renormalize:
        w = w / w.sum(dim=-1, keepdim=True)
    return w, ids.to(torch.int32)


def fused_moe(x: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, topk_w: torch.Tensor, topk_ids: torch.Tensor,
              w13_scale: Optional[torch.Tensor] = None, w2_scale: Optional[torch.Tensor] = None,
              block_scales: bool = False, activation: str = "silu") -> torch.Tensor:
    """out[t] = sum_j topk_w[t, j] * W2_e( act(Wg_e x_t) * (Wu_e x_t) ), e = topk_ids[t, j];
    negative ids (expert on another rank) contribute nothing.  fp32 accumulation.  ``block_scales``:
    the weight scales are per 128 x 128 block instead of per expert."""
    T, H = x.shape
    E, I2, _ = w13.shape
    I = I2 // 2
    out = torch.zeros((T, w2.shape[1]), dtype=torch.float32, device=x.device)
    xf = x.float()
    for e in range(E):
        tok, slot = torch.where(topk_ids == e)
        if tok.numel() == 0:
            continue
        if block_scales:
            a = w13[e].float() * expand_block_scales(w13_scale[e].float(), I2, H)
            b = w2[e].float() * expand_block_scales(w2_scale[e].float(), w2.shape[1], I)
        else:
            a = w13[e].float() * (w13_scale[e].float() if w13_scale is not None else 1.0)
            b = w2[e].float() * (w2_scale[e].float() if w2_scale is not None else 1.0)
        xe = xf[tok]
        if block_scales:      # what the device kernels do: activations rounded to e4m3 per group of 128
            q, s_ = per_group_fp8_quant(xe.to(x.dtype))
            xe = (q.float().view(*s_.shape, -1) * s_.unsqueeze(-1)).view(xe.shape)
        h = xe @ a.t()
        act = (_act(activation, h[:, :I]) * h[:, I:]).to(x.dtype).float()
        if block_scales:
            q, s_ = per_group_fp8_quant(act.to(x.dtype))
            act = (q.float().view(*s_.shape, -1) * s_.unsqueeze(-1)).view(act.shape)
        out.index_add_(0, tok, (act @ b.t()) * topk_w[tok, slot].float().unsqueeze(1))
    return out.to(x.dtype)


def int8_quant(x, scale=None):
    xf = x.float()
    if scale is None:
        s = (xf.abs().amax(-1, keepdim=True) / 127.0).clamp(min=1e-10)
    else:
        s = scale.float().reshape(())
    q = torch.round(xf / s).clamp(-128, 127).to(torch.int8)
    return q, (s if scale is None else scale)


def int8_quant_azp(x, scale=None, azp=None):
    xf = x.float()
    if scale is None:
        mx, mn = xf.amax(-1, keepdim=True), xf.amin(-1, keepdim=True)
        s = ((mx - mn) / 255.0).clamp(min=1e-10)
        z = torch.round(-128.0 - mn / s).to(torch.int32)
        q = (torch.round(xf / s) + z).clamp(-128, 127).to(torch.int8)
        return q, s, z
    s = scale.float().reshape(())
    z = azp.reshape(()).to(torch.int32) if azp is not None else torch.zeros((), dtype=torch.int32, device=x.device)
    q = (torch.round(xf / s) + z).clamp(-128, 127).to(torch.int8)
    return q, scale, azp


# ---- state-space layers ---------------------------------------------------------------------------
def causal_conv1d(x, weight, bias, conv_state, qsl, seq_lens, block_table, activation=True):
    T, C = x.shape
    W = weight.shape[1]
    out = torch.zeros((T, C), dtype=x.dtype, device=x.device)
    w = weight.float()
    for s in range(seq_lens.shape[0]):
        sl = int(seq_lens[s])
        if sl <= 0:
            continue
        t0, t1 = int(qsl[s]), int(qsl[s + 1])
        L = t1 - t0
        slot = int(block_table[s, 0])
        prev = conv_state[slot].float() if sl - L > 0 else torch.zeros((C, W - 1), device=x.device)
        seq = torch.cat([prev.t(), x[t0:t1].float()], dim=0)              # [W-1+L, C]
        win = seq.unfold(0, W, 1)                                           # [L, C, W]
        y = (win * w.unsqueeze(0)).sum(-1)
        if bias is not None:
            y = y + bias.float()
        if activation:
            y = torch.nn.functional.silu(y)
        out[t0:t1] = y.to(x.dtype)
        conv_state[slot] = seq[-(W - 1):].t().to(conv_state.dtype)
    return out


def selective_scan(u, delta, A, B, C, D, z, delta_bias, ssm_state, qsl, seq_lens, block_table, softplus=True,
                   num_groups=1):
    T, Ch = u.shape
    Nst = ssm_state.shape[2]
    E = delta.shape[1]
    rep = Ch // E
    out = torch.zeros((T, Ch), dtype=u.dtype, device=u.device)
    A_full = (A.float().repeat_interleave(rep).unsqueeze(1).expand(Ch, Nst) if A.dim() == 1 else A.float())
    Dv = D.float().repeat_interleave(rep) if D is not None else None
    gch = Ch // num_groups
    for s in range(seq_lens.shape[0]):
        sl = int(seq_lens[s])
        if sl <= 0:
            continue
        t0, t1 = int(qsl[s]), int(qsl[s + 1])
        slot = int(block_table[s, 0])
        h = ssm_state[slot].float().clone() if sl - (t1 - t0) > 0 else torch.zeros((Ch, Nst), device=u.device)
        for t in range(t0, t1):
            dt = delta[t].float()
            if delta_bias is not None:
                dt = dt + delta_bias.float()
            if softplus:
                dt = torch.nn.functional.softplus(dt)
            dt = dt.repeat_interleave(rep)                                   # [Ch]
            x = u[t].float()
            Bt = B[t].float().view(num_groups, Nst).repeat_interleave(gch, dim=0)   # [Ch, N]
            Ct = C[t].float().view(num_groups, Nst).repeat_interleave(gch, dim=0)
            h = torch.exp(dt.unsqueeze(1) * A_full) * h + (dt * x).unsqueeze(1) * Bt
            y = (h * Ct).sum(-1)
            if Dv is not None:
                y = y + Dv * x
            if z is not None:
                y = y * torch.nn.functional.silu(z[t].float())
            out[t] = y.to(u.dtype)
        ssm_state[slot] = h
    return out


# ---- 4-bit weights -----------------------------------------------------------------------------------
def w4_unpack(qw: torch.Tensor) -> torch.Tensor:
    """int32 [N, K/8] in the kernel's nibble order -> int64 [N, K] of values 0..15."""
    w = qw.to(torch.int64) & 0xFFFFFFFF
    cols = [(w >> (4 * (e // 2 + 4 * (e % 2)))) & 0xF for e in range(8)]
    return torch.stack(cols, dim=-1).reshape(qw.shape[0], -1)


def cb4_dequant(codes, scales, table, group, hi_first, dtype):
    N_, K2 = codes.shape
    first, second = (codes >> 4, codes & 15) if hi_first else (codes & 15, codes >> 4)
    idx = torch.stack((first, second), dim=-1).reshape(N_, 2 * K2).long()
    vals = table.float()[idx].view(N_, -1, group) * scales.float().unsqueeze(-1)
    return vals.view(N_, 2 * K2).to(dtype)


def w4_dequant(qw, scales, zeros, group, dtype):
    q = w4_unpack(qw).float()
    s = scales.float().repeat_interleave(group, dim=1)
    z = zeros.float().repeat_interleave(group, dim=1)
    return ((q - z) * s).to(dtype)


# ---- multi-LoRA --------------------------------------------------------------------------------------
def lora_update(y, x, A, B, tok_slot):
    for s in torch.unique(tok_slot).tolist():
        if s < 0:
            continue
        idx = torch.nonzero(tok_slot == s).squeeze(1)
        d = (x[idx].float() @ A[s].float().t()) @ B[s].float().t()
        y[idx] = (y[idx].float() + d).to(y.dtype)


# ---- multi-head latent attention -----------------------------------------------------------------------
def mla_cache_write(kv_c, k_pe, kc, vt, slot_mapping, kv_scale=1.0):
    BS = kc.shape[1]
    Rk = kv_c.shape[1]
    if kc.dtype == torch.float8_e4m3fn:
        kv_c, k_pe = kv_c.float() / kv_scale, k_pe.float() / kv_scale
    for t in range(kv_c.shape[0]):
        slot = int(slot_mapping[t])
        if slot < 0:
            continue
        b, o = slot // BS, slot % BS
        kc[b, o, :Rk] = kv_c[t].to(kc.dtype)
        kc[b, o, Rk:] = k_pe[t].to(kc.dtype)
        vt[b, :, o] = kv_c[t].to(vt.dtype)


def mla_attention(q, kc, vt, block_table, seq_lens, q_start_loc, scale, kv_scale=1.0):
    T, H, DK = q.shape
    DV, BS = vt.shape[1], kc.shape[1]
    if kc.dtype != torch.float8_e4m3fn:
        kv_scale = 1.0
    out = torch.zeros((T, H, DV), dtype=q.dtype, device=q.device)
    for s in range(seq_lens.shape[0]):
        L = int(seq_lens[s])
        a, b = int(q_start_loc[s]), int(q_start_loc[s + 1])
        if L <= 0 or b <= a:
            continue
        nblk = (L + BS - 1) // BS
        blocks = block_table[s, :nblk].long()
        K = kc[blocks].reshape(-1, DK)[:L].float() * kv_scale              # [L, DK]
        V = vt[blocks].permute(0, 2, 1).reshape(-1, DV)[:L].float() * kv_scale   # [L, DV]
        qs = q[a:b].float()                                                # [n, H, DK]
        n = b - a
        sc = torch.einsum("nhd,ld->nhl", qs, K) * scale
        pos = torch.arange(L, device=q.device).view(1, 1, L)
        lim = (L - n + torch.arange(n, device=q.device) + 1).view(n, 1, 1)
        sc = sc.masked_fill(pos >= lim, float("-inf"))
        out[a:b] = torch.einsum("nhl,ld->nhd", torch.softmax(sc, dim=-1), V).to(q.dtype)
    return out
