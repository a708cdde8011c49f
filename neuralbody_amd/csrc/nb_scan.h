// Internal: device-wide exclusive scan (nb_scan.hip).
#pragma once
#include "nb_common.h"

long long nb_scan_blocks(long long n);
// out[i] = sum_{k<i} flags[k]; *total = sum of all flags.  block_sums: nb_scan_blocks(n) ints.
int nb_exclusive_scan(const int *flags, int *out, int *total, long long n, int *block_sums, hipStream_t st);
// split a scratch buffer of nb_scan_scratch_size(n) bytes into [flags n][pos n][block sums]
void nb_scan_carve(void *scratch, long long n, int **flags, int **pos, int **block_sums);
