/* A C host of libhmmsort_hip.so: the calls a compiled (non-Python) host makes on this path, with plain
 * pointers only.  Reads a signal and a K x N template matrix from a binary file written by the test,
 * builds the state space with the library's host helpers (types.jl:65-127), decodes (viterbi.jl:44),
 * runs one EM step (baumwelch.jl:362) and writes path, ll, mu, sigma, lp for the test to compare with
 * what the Python binding gets for the same inputs.
 *   gcc -O2 -I include tests/c_host/host.c -o host -L hmmspikesorter.jl_amd -lhmmsort_hip -Wl,-rpath,...
 *   ./host in.bin out.bin */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "hmmsort.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != 0) {                                                              \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, hmmsort_last_error());     \
            return 2;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int64_t hdr[3]; /* N, K, T */
    double sigma, lp_in[16];
    if (fread(hdr, sizeof(int64_t), 3, f) != 3 || fread(&sigma, sizeof(double), 1, f) != 1) return 1;
    const int64_t N = hdr[0], K = hdr[1], T = hdr[2];
    if (N < 1 || N > 16 || fread(lp_in, sizeof(double), (size_t)N, f) != (size_t)N) return 1;
    double *mu = malloc(sizeof(double) * (size_t)(K * N)), *y = malloc(sizeof(double) * (size_t)T);
    if (fread(mu, sizeof(double), (size_t)(K * N), f) != (size_t)(K * N) || fread(y, sizeof(double), (size_t)T, f) != (size_t)T)
        return 1;
    fclose(f);

    /* StateMatrix(N, K, lp, allow_overlaps=false): state table and transition list */
    const int64_t S = hmmsort_generate_states(N, K, 0, NULL);
    if (S < 0) { fprintf(stderr, "%s\n", hmmsort_last_error()); return 2; }
    int16_t *states = malloc(sizeof(int16_t) * (size_t)(N * S));
    hmmsort_generate_states(N, K, 0, states);
    const int64_t R = hmmsort_build_transitions(N, K, lp_in, N, 0, NULL, 0);
    if (R < 0) { fprintf(stderr, "%s\n", hmmsort_last_error()); return 2; }
    hmm_trans *tr = malloc(sizeof(hmm_trans) * (size_t)R);
    hmmsort_build_transitions(N, K, lp_in, N, 0, tr, R);

    int16_t *x = malloc(sizeof(int16_t) * (size_t)T);
    double ll = 0.0, sigma_new = 0.0;
    CHECK(hmmsort_viterbi(y, T, states, N, K, S, tr, R, mu, sigma, x, &ll));
    double *lp = malloc(sizeof(double) * (size_t)R), *pp = malloc(sizeof(double) * (size_t)S);
    int64_t nlp = 0;
    CHECK(hmmsort_em_step(y, T, states, N, K, S, tr, R, mu, sigma, &sigma_new, lp, R, &nlp, pp));
    CHECK(hmmsort_shutdown());

    f = fopen(argv[2], "wb");
    if (!f) return 1;
    fwrite(&S, sizeof(int64_t), 1, f);
    fwrite(&nlp, sizeof(int64_t), 1, f);
    fwrite(&ll, sizeof(double), 1, f);
    fwrite(&sigma_new, sizeof(double), 1, f);
    fwrite(x, sizeof(int16_t), (size_t)T, f);
    fwrite(mu, sizeof(double), (size_t)(K * N), f);      /* overwritten in place, as update() does */
    fwrite(lp, sizeof(double), (size_t)nlp, f);
    fclose(f);
    printf("C host: S=%lld R=%lld ll=%.6f sigma=%.9f\n", (long long)S, (long long)R, ll, sigma_new);
    return 0;
}
