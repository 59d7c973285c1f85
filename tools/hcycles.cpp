// hcycles.cpp -- short-cycle census of an LDPC parity-check matrix (the checker half of SURVEY.md 8(f)-3; the generator
// half is tools/hgen.cpp).
//
// Behaviour of the reference's checkers, re-implemented on CSR:
//   Matlab/Hcyclefinder.m:60-147            tree rooted at every variable node t: tier-1 variable nodes (the other
//                                            neighbours of t's checks), check tier 2 (their other checks), variable tier 2;
//                                            a repeated node in a tier closes a cycle through the root -- repeated
//                                            variable in tier 1: 4-cycle (:86-95), repeated check in check tier 2:
//                                            6-cycle (:110-121), repeated variable in variable tier 2: 8-cycle (:135-143).
//                                            Counts are "adjacent equal pairs of the sorted tier", per root, like the script.
//   Matlab/Cycle_Finder_length4_fromroot.m  the yes/no form of the 4-cycle test used while a matrix is being built
//   Matlab/Cycle_Finder_length6.m:20-70     the yes/no form of the 6-cycle test (a root with a 4-cycle counts as having one)
//
// Input: the CSR fixture format "LDPCCSR1" (tools/export_reference_data.py, tools/hgen.cpp).
//   g++ -O2 -std=c++17 -o tools/bin/hcycles tools/hcycles.cpp
//   tools/bin/hcycles ldpc_erasure_codes_amd/data/code_n2040_k1530.csr.bin [--roots] [--no8]
// Output (one line, key=value): n k m nnz roots4 roots6 roots8 pairs4 pairs6 pairs8 girth_at_least; with --roots a
// second line "roots6: i j ..." lists the 0-based variable roots that see a 6-cycle.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: hcycles <code.csr.bin> [--roots] [--no8]\n");
        return 2;
    }
    bool list_roots = false, do8 = true;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--roots")) list_roots = true;
        else if (!strcmp(argv[i], "--no8")) do8 = false;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 3; }
    char magic[8];
    uint32_t hdr[4];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "LDPCCSR1", 8) || fread(hdr, 4, 4, f) != 4) {
        fprintf(stderr, "%s: not an LDPCCSR1 file\n", argv[1]);
        return 3;
    }
    const int n = (int)hdr[0], k = (int)hdr[1], m = (int)hdr[2];
    const size_t nnz = hdr[3];
    std::vector<uint32_t> row_ptr(m + 1);
    std::vector<uint16_t> cols(nnz);
    if (fread(row_ptr.data(), 4, m + 1, f) != (size_t)m + 1 || fread(cols.data(), 2, nnz, f) != nnz || row_ptr[m] != nnz || m != n - k) {
        fprintf(stderr, "%s: inconsistent file\n", argv[1]);
        return 3;
    }
    fclose(f);
    // Clist: checks of every variable node (Hcyclefinder.m:46-56), ascending
    std::vector<uint32_t> cptr(n + 1, 0);
    for (size_t e = 0; e < nnz; e++) cptr[cols[e] + 1]++;
    for (int j = 0; j < n; j++) cptr[j + 1] += cptr[j];
    std::vector<int> chk(nnz);
    {
        std::vector<uint32_t> fill(cptr.begin(), cptr.end() - 1);
        for (int r = 0; r < m; r++)
            for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) chk[fill[cols[e]]++] = r;
    }
    auto pairs = [](std::vector<int> &v) {   // adjacent equal pairs of the sorted list (Hcyclefinder.m:86-88)
        std::sort(v.begin(), v.end());
        long c = 0;
        for (size_t i = 0; i + 1 < v.size(); i++) c += v[i] == v[i + 1];
        return c;
    };
    long roots4 = 0, roots6 = 0, roots8 = 0, pairs4 = 0, pairs6 = 0, pairs8 = 0;
    std::vector<int> r6;
    std::vector<int> v1, v1par, c2, c2par, v2, tmp;
    for (int t = 0; t < n; t++) {
        v1.clear(); v1par.clear();
        for (uint32_t a = cptr[t]; a < cptr[t + 1]; a++) {          // check tier 1
            const int c = chk[a];
            for (uint32_t e = row_ptr[c]; e < row_ptr[c + 1]; e++)
                if (cols[e] != t) { v1.push_back(cols[e]); v1par.push_back(c); }   // :70-80
        }
        tmp = v1;
        const long p4 = pairs(tmp);                                  // :86-95
        c2.clear(); c2par.clear();
        for (size_t i = 0; i < v1.size(); i++)                       // :99-108: the other checks of every tier-1 variable
            for (uint32_t a = cptr[v1[i]]; a < cptr[v1[i] + 1]; a++)
                if (chk[a] != v1par[i]) { c2.push_back(chk[a]); c2par.push_back(v1[i]); }
        tmp = c2;
        const long p6 = pairs(tmp);                                  // :110-121
        long p8 = 0;
        if (do8) {
            v2.clear();
            for (size_t i = 0; i < c2.size(); i++)                   // :124-133
                for (uint32_t e = row_ptr[c2[i]]; e < row_ptr[c2[i] + 1]; e++)
                    if (cols[e] != c2par[i]) v2.push_back(cols[e]);
            p8 = pairs(v2);                                          // :135-143
        }
        pairs4 += p4; pairs6 += p6; pairs8 += p8;
        roots4 += p4 > 0; roots8 += p8 > 0;
        if (p6 > 0 || p4 > 0) { roots6++; r6.push_back(t); }          // Cycle_Finder_length6.m:66-69: a 4-cycle counts too
    }
    const int girth = roots4 ? 4 : (roots6 ? 6 : ((do8 && roots8) ? 8 : (do8 ? 10 : 8)));
    printf("n=%d k=%d m=%d nnz=%zu roots4=%ld roots6=%ld roots8=%ld pairs4=%ld pairs6=%ld pairs8=%ld girth_at_least=%d\n", n, k, m, nnz,
           roots4, roots6, roots8, pairs4, pairs6, pairs8, girth);
    if (list_roots) {
        printf("roots6:");
        for (int t : r6) printf(" %d", t);
        printf("\n");
    }
    return 0;
}
