// Shared between the attention translation units: kernel parameter block and the bare 2^x.
#pragma once
#include "common.h"

// 2^x as the bare v_exp_f32 instruction.  fexp2() wraps it in a range check and a rescale for results
// below the smallest normal float (seven instructions instead of one); softmax terms that small are
// zero for every purpose here.
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnParams {
  const void* q;            // [T, Hq, D]
  const void* k_cache;
  const void* v_cache;
  void* out;                // [T, Hq, D]
  float* part_out;          // [T, Hq, nsplit, D]   (split mode only)
  float* part_lse;          // [T, Hq, nsplit]      log2-domain log-sum-exp
  const int* block_table;   // [S, max_blocks]
  const int* seq_lens;      // [S] keys visible to the LAST query token of the sequence
  const int* q_start_loc;   // [S+1]
  const float* alibi;       // [Hq] or nullptr
  int num_seqs, Hq, Hkv, max_blocks, nsplit;
  long q_stride, out_stride;   // elements between consecutive tokens
  float scale, softcap;    // scale already includes the K-cache scale of an fp8 cache
  float v_scale;            // V-cache scale (1 for 16-bit caches)
  int window;               // sliding window size, 0 = none
  int q_tile;               // query tokens per workgroup (prefill form)
  int causal;               // 1: query token i sees keys up to its own position; 0: every key of the sequence
  int num_tokens;           // T (row length of lse_out)
  float* lse_out;           // [Hq, T] natural-log log-sum-exp per (head, token), or nullptr (prefill form)
  int max_q_len;            // longest query run of the batch when the host knows it, else 0 (wide prefill form)
};

// Key selection of the sparse prefill form: per (sequence, head) a list of single key positions ("vertical" lines of
// the score matrix) and a list of query-minus-key distances ("slash" lines), both in absolute positions of the
// request; key_base / q_base place the sequence's table and its first query row on that axis.
struct SparseSel {
  const int* vert;      // [S, Hq, NV]
  const int* vcount;    // [S, Hq]
  const int* slash;     // [S, Hq, NS]
  const int* scount;    // [S, Hq]
  const int* key_base;  // [S]
  const int* q_base;    // [S]
  int NV, NS;
};

// Wide prefill form (attention_prefill.hip): 512 threads, 256 query columns per workgroup, head size 128.
// Returns -100 when the shape is outside what that kernel serves (the caller then takes the 4-wave form).
int vda_launch_prefill_wide(const AttnParams& p, int block_size, int dtype, int kv_fp8, hipStream_t st);
