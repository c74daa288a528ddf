// Prompt-lookup ("n-gram") draft proposer: find the most recent earlier occurrence of the
// sequence's last n tokens (longest n in [min_n, max_n] first) and propose the k tokens that
// followed it.  Runs on the host inside the decode loop for every request, hence native.
// Counterpart of the reference's numba kernel in vllm/v1/spec_decode/ngram_proposer.py.
#include <cstdint>

extern "C" __attribute__((visibility("default")))
int vda_ngram_propose(const int32_t* tokens, int64_t n_tokens, int min_n, int max_n, int k, int32_t* out) {
    if (k <= 0 || n_tokens < 2) return 0;
    for (int n = max_n; n >= min_n; --n) {
        if (n <= 0 || n_tokens <= n) continue;
        const int32_t* pat = tokens + (n_tokens - n);
        // candidate start positions s: tokens[s .. s+n) == pat, and at least one token follows
        for (int64_t s = n_tokens - n - 1; s >= 0; --s) {
            int j = n - 1;
            while (j >= 0 && tokens[s + j] == pat[j]) --j;
            if (j >= 0) continue;
            int64_t start = s + n;
            int64_t avail = n_tokens - start;
            int cnt = avail < k ? int(avail) : k;
            for (int i = 0; i < cnt; ++i) out[i] = tokens[start + i];
            return cnt;
        }
    }
    return 0;
}

// Batched form: rows of a [n_rows, stride] token matrix with per-row lengths; out is [n_rows, k],
// counts[n_rows].  Rows with do_row == 0 are skipped.
extern "C" __attribute__((visibility("default")))
void vda_ngram_propose_batch(const int32_t* tokens, int64_t stride, const int64_t* row_ids, const int64_t* lens,
                             const uint8_t* do_row, int n_rows, int min_n, int max_n, int k, int32_t* out,
                             int32_t* counts) {
    for (int r = 0; r < n_rows; ++r) {
        counts[r] = do_row[r] ? vda_ngram_propose(tokens + row_ids[r] * stride, lens[r], min_n, max_n, k,
                                                  out + int64_t(r) * k)
                              : 0;
    }
}
