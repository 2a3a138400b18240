/* Checks that the 12x12 circulant Rescue matrix of the parameter blob is MDS over Goldilocks:
 * every square submatrix is non-singular (all C(24,12)-1 = 2 704 155 minors non-zero).
 * Test infrastructure (oracle/); usage: mds_check <params_default.bin>   -> prints "MDS ok <count>". */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
#define P 0xffffffff00000001ULL
static uint64_t mulm(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % P); }
static uint64_t powm(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mulm(r, a);
        a = mulm(a, a);
        e >>= 1;
    }
    return r;
}
static uint64_t M[12][12];

static int nonsingular(const int *rows, const int *cols, int k) {
    uint64_t a[12][12];
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) a[i][j] = M[rows[i]][cols[j]];
    for (int c = 0; c < k; c++) {
        int piv = -1;
        for (int r = c; r < k; r++)
            if (a[r][c]) {
                piv = r;
                break;
            }
        if (piv < 0) return 0;
        if (piv != c)
            for (int j = 0; j < k; j++) {
                uint64_t t = a[c][j];
                a[c][j] = a[piv][j];
                a[piv][j] = t;
            }
        uint64_t inv = powm(a[c][c], P - 2);
        for (int r = c + 1; r < k; r++) {
            if (!a[r][c]) continue;
            uint64_t f = mulm(a[r][c], inv);
            for (int j = c; j < k; j++) a[r][j] = (a[r][j] + P - mulm(f, a[c][j])) % P;
        }
    }
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint8_t blob[2816];
    if (fread(blob, 1, sizeof blob, f) != sizeof blob) return 2;
    fclose(f);
    memcpy(M, blob + 32, sizeof M);
    long count = 0;
    for (unsigned rm = 1; rm < 4096; rm++) {
        int rows[12], k = 0;
        for (int i = 0; i < 12; i++)
            if (rm >> i & 1) rows[k++] = i;
        for (unsigned cm = 1; cm < 4096; cm++) {
            if (__builtin_popcount(cm) != k) continue;
            int cols[12], c = 0;
            for (int i = 0; i < 12; i++)
                if (cm >> i & 1) cols[c++] = i;
            if (!nonsingular(rows, cols, k)) {
                printf("singular minor rows=%03x cols=%03x\n", rm, cm);
                return 1;
            }
            count++;
        }
    }
    printf("MDS ok %ld\n", count);
    return 0;
}
