/*
 * c_abi_demo.c -- libcutadapt_hip.so from plain C, without Python or PyTorch.
 *
 * What a non-Python host (or a cutadapt maintainer writing a C extension) does:
 *   1. describe the adapter exactly like cutadapt's BackAdapter would configure its Aligner and
 *      KmerFinder (reference adapters.py:810-832; the k-mer search sets below are what
 *      kmer_heuristic.create_positions_and_kmers returns for this adapter, e = 0.1, overlap 3),
 *   2. hand packed reads over (host pointers here; the d_* entry points take HBM pointers),
 *   3. read back (ref_start, ref_stop, query_start, query_stop, score, errors) per read.
 *
 * Build:  gcc -std=c11 -O2 examples/c_abi_demo.c -Iinclude -Lcutadapt_amd -lcutadapt_hip \
 *             -Wl,-rpath,$PWD/cutadapt_amd -Wl,-rpath,/opt/rocm/lib -o c_abi_demo
 * Prints one line per read and exits 0 when the results are the expected ones
 * (the same numbers `cutadapt -a AGATCGGAAGAGCACACGTCTGAACTCCAGTCA` reports).
 */
#include <stdio.h>
#include <string.h>

#include "cutadapt_hip.h"

#define ADAPTER "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"

int main(void) {
    char msg[512];
    int n_dev = 0;
    if (cah_device_count(&n_dev) != CAH_OK || n_dev < 1) {
        fprintf(stderr, "no HIP device: this library has no CPU fallback\n");
        return 2;
    }

    /* the prefilter of BackAdapter(ADAPTER, max_errors=0.1, min_overlap=3) */
    static const char *k3[] = {"AGA"}, *k4[] = {"AGAT"}, *k19[] = {"AGATC", "GGAAG"};
    static const char *k29[] = {"AAGAGCA", "AGATCGG", "CACGTC"};
    static const char *k33[] = {"AGATCGGA", "CGTCTGA", "ACTCCAG", "AGAGCACA"};
    static const char *kall[] = {"TCCAGTCA", "AGATCGGAA", "GTCTGAAC", "GAGCACAC"};
    const cah_kmer_set sets[] = {
        {-3, 0, k3, 1}, {-4, 0, k4, 1}, {-19, 0, k19, 2}, {-29, 0, k29, 3}, {-33, 0, k33, 4}, {0, 0, kall, 4},
    };
    cah_adapter_desc d;
    memset(&d, 0, sizeof(d));
    d.sequence = ADAPTER;
    d.length = (int32_t)strlen(ADAPTER);
    d.max_error_rate = 0.1;
    d.flags = 2 | 4 | 8;            /* QUERY_START | REFERENCE_END | QUERY_STOP = Where.BACK */
    d.indel_cost = 1;
    d.min_overlap = 3;
    d.kind = CAH_KIND_ALIGNER;
    d.kmer_sets = sets;
    d.n_kmer_sets = 6;

    cah_plan *plan = NULL;
    if (cah_plan_create(&d, 1, &plan) != CAH_OK) {
        cah_last_error(msg, sizeof(msg));
        fprintf(stderr, "cah_plan_create: %s\n", msg);
        return 1;
    }

    /* four reads, packed back to back */
    const char *reads[] = {
        "GGCTTACGATCCGATAGATCGGAAGAGCACACGTCTGAACTCCAGTCACTTAGGC",   /* full adapter at 15 */
        "ACGTTGCATGCCATGGATCGATCGTAGCTAGCTAGGATCGATCGATCGATGCATG",   /* no adapter */
        "TTGACCGATAGCATCGACTAGCATCGAGATCGGAAGAGCACTCGTCTGAACTCCA",   /* 1 mismatch, truncated */
        "CCGATAGCATGCATGCAGCTAGCTAGCATCGATCGATCGATGCATCGATCAGATC",   /* 5-base overlap at the end */
    };
    const int n = 4;
    uint8_t seqs[1024];
    int64_t offsets[5];
    int64_t pos = 0;
    for (int i = 0; i < n; i++) {
        offsets[i] = pos;
        memcpy(seqs + pos, reads[i], strlen(reads[i]));
        pos += (int64_t)strlen(reads[i]);
    }
    offsets[n] = pos;

    int32_t out6[4][6];
    int32_t best[4];
    uint8_t status[4];
    if (cah_match_batch_host(plan, seqs, offsets, n, &out6[0][0], best, status) != CAH_OK) {
        cah_last_error(msg, sizeof(msg));
        fprintf(stderr, "cah_match_batch_host: %s\n", msg);
        return 1;
    }
    const int expect_status[4] = {CAH_MATCH, CAH_NONE, CAH_MATCH, CAH_MATCH};
    const int expect_qstart[4] = {15, 0, 26, 50};
    const int expect_errors[4] = {0, 0, 1, 0};
    int ok = 1;
    for (int i = 0; i < n; i++) {
        if (status[i] == CAH_MATCH)
            printf("read %d: adapter[%d:%d] matches read[%d:%d], score %d, %d error(s) -> keep %d bases\n", i,
                   out6[i][0], out6[i][1], out6[i][2], out6[i][3], out6[i][4], out6[i][5], out6[i][2]);
        else
            printf("read %d: no adapter\n", i);
        ok &= status[i] == expect_status[i];
        if (status[i] == CAH_MATCH) ok &= out6[i][2] == expect_qstart[i] && out6[i][5] == expect_errors[i];
    }
    cah_plan_destroy(plan);
    printf(ok ? "OK\n" : "UNEXPECTED RESULT\n");
    return ok ? 0 : 1;
}
