/* A plain C caller of libseekr_hip.so (no Python anywhere): the reference's three-line recipe
 *     counts = BasicCounter(fasta, k=K).make_count_file()        (kmer_counts.py:243-262)
 *     r      = pearson(counts, counts)                           (pearson.py:32-44)
 * through the one-call host forms of include/seekr_hip.h, results written as .npy with the
 * library's own writer so that the test can load them.
 *
 *   host_example <in.fasta> <k> <counts.npy> <pearson.npy>
 */
#include <stdio.h>
#include <stdlib.h>

#include "seekr_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != SKR_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, skr_last_error());     \
            return 1 - rc_; /* exit code = 1 + |status| */                       \
        }                                                                        \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 5) {
        fprintf(stderr, "usage: %s in.fasta k counts.npy pearson.npy\n", argv[0]);
        return 64;
    }
    const int k = atoi(argv[2]);
    skr_ctx* ctx = NULL;
    skr_seqs* seqs = NULL;
    int64_t n = 0, bases = 0, longest = 0;
    CHECK(skr_ctx_create(0, &ctx));
    CHECK(skr_seqs_from_fasta(ctx, argv[1], "AGTC", &seqs));
    CHECK(skr_seqs_info(seqs, &n, &bases, &longest));
    const int64_t cols = (int64_t)1 << (2 * k);
    float* counts = malloc((size_t)n * cols * sizeof(float));
    float* mean = malloc((size_t)cols * sizeof(float));
    float* std = malloc((size_t)cols * sizeof(float));
    float* r = malloc((size_t)n * n * sizeof(float));
    if (!counts || !mean || !std || !r) return 65;
    int has_nan = 0;
    CHECK(skr_host_get_counts(ctx, seqs, k, SKR_LOG2_POST, 1, NULL, SKR_F32, 1, NULL, SKR_F32, counts, mean, std, &has_nan));
    CHECK(skr_host_pearson(ctx, counts, n, counts, n, cols, SKR_F32, 1, SKR_PREC_F16X3, r));
    CHECK(skr_host_save_npy(counts, SKR_F32, n, cols, 0, argv[3]));
    CHECK(skr_host_save_npy(r, SKR_F32, n, n, 0, argv[4]));
    printf("%lld sequences, %lld bases, %lld columns, nan=%d, mean[0]=%.9g std[0]=%.9g\n", (long long)n, (long long)bases,
           (long long)cols, has_nan, (double)mean[0], (double)std[0]);
    free(counts);
    free(mean);
    free(std);
    free(r);
    CHECK(skr_seqs_free(seqs));
    CHECK(skr_ctx_destroy(ctx));
    return 0;
}
