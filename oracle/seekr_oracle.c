/* C restatement of the counting / column-sum part of the oracle.  TEST INFRASTRUCTURE ONLY:
 * linked by nothing in seekr_amd/; used by tests/ (full-size checks where the numpy/Python oracle
 * is too slow) and by bench.py's cpu_baseline leg.  Parity status: pinned — checked against the
 * numpy oracle and the golden vectors in tests/test_oracle_golden.py.
 *
 * Reference lines restated (CalabreseLab/seekr v2.0.2):
 *   orc_count_u32      kmer_counts.py:142-150  (window index, first base most significant;
 *                                               windows holding a non-alphabet byte are skipped)
 *   orc_per_kb_f32     kmer_counts.py:144-147,150 (n float64 additions of 1000/W, stored as float32)
 *   orc_colsum_seq_f32 kmer_counts.py:168,174  (numpy axis-0 reduce: rows added in index order, float32)
 */
#include <stdint.h>
#include <string.h>

int orc_count_u32(const unsigned char* bases, const int64_t* offsets, int64_t n, int k,
                  const unsigned char* alphabet, int alpha_len, uint32_t* out) {
    int code[256];
    for (int i = 0; i < 256; i++) code[i] = -1;
    for (int c = 0; c < alpha_len; c++) code[alphabet[c]] = c;
    int64_t ncols = 1;
    for (int i = 0; i < k; i++) ncols *= alpha_len;
    for (int64_t s = 0; s < n; s++) {
        const unsigned char* seq = bases + offsets[s];
        const int64_t len = offsets[s + 1] - offsets[s];
        uint32_t* row = out + s * ncols;
        memset(row, 0, (size_t)ncols * sizeof(uint32_t));
        for (int64_t w = 0; w + k <= len; w++) {
            int64_t idx = 0;
            int ok = 1;
            for (int p = 0; p < k; p++) {
                const int c = code[seq[w + p]];
                if (c < 0) { ok = 0; break; }
                idx = idx * alpha_len + c;
            }
            if (ok) row[idx]++;
        }
    }
    return 0;
}

/* returns -1 if a sequence has length k-1 (ZeroDivisionError in the reference) */
int orc_per_kb_f32(const uint32_t* counts, const int64_t* lengths, int64_t n, int k, int64_t ncols, float* out) {
    for (int64_t s = 0; s < n; s++) {
        const int64_t w = lengths[s] - k + 1;
        if (w == 0) return -1;
        const uint32_t* row = counts + s * ncols;
        float* dst = out + s * ncols;
        if (w < 0) {
            memset(dst, 0, (size_t)ncols * sizeof(float));
            continue;
        }
        const double inc = 1000.0 / (double)w;
        for (int64_t j = 0; j < ncols; j++) {
            double acc = 0.0;
            for (uint32_t t = 0; t < row[j]; t++) acc += inc;
            dst[j] = (float)acc;
        }
    }
    return 0;
}

/* acc[j] = fl32(acc[j] + x[i][j]) for i = 0..rows-1 (compile WITHOUT -ffast-math) */
void orc_colsum_seq_f32(const float* x, int64_t rows, int64_t cols, float* acc) {
    for (int64_t i = 0; i < rows; i++) {
        const float* r = x + i * cols;
        for (int64_t j = 0; j < cols; j++) acc[j] = acc[j] + r[j];
    }
}
