// ref_edits_driver.cpp -- thin command-line driver around the REFERENCE's own Edit class (src/Edits.cpp, include/Edits.h:
// Edit::optimizeEditScript, Edits::applyEdits), compiled where it lies.  TEST INFRASTRUCTURE ONLY (see oracle/README.md);
// nothing here restates the algorithm, it only feeds scripts in and dumps what the reference makes of them.
//
// in : u32 n_cases | per case: u32 n_ops, u32 orig_len, orig bytes, then per op: u8 type (0 SAME 1 INSERT 2 DELETE), u8 base, u32 num
// out: per case: u64 editDis | u32 n_new | per new op: u8 type, u8 base, u32 num | u32 len + bytes of applyEdits(orig, raw)
//      | u32 len + bytes of applyEdits(orig, optimised)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iterator>
#include <string>
#include <vector>
#include "Edits.h"

template <class T> static void rd(FILE *f, T *p, size_t cnt) { if (cnt && fread(p, sizeof(T), cnt, f) != cnt) { fprintf(stderr, "short read\n"); exit(2); } }
template <class T> static void wr(FILE *f, const T *p, size_t cnt) { if (cnt && fwrite(p, sizeof(T), cnt, f) != cnt) { fprintf(stderr, "short write\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: nsref_edits in.bin out.bin\n"); return 1; }
    FILE *fi = fopen(argv[1], "rb"), *fo = fopen(argv[2], "wb");
    if (!fi || !fo) { perror("open"); return 1; }
    uint32_t n_cases;
    rd(fi, &n_cases, 1);
    for (uint32_t c = 0; c < n_cases; ++c) {
        uint32_t n_ops, olen;
        rd(fi, &n_ops, 1);
        rd(fi, &olen, 1);
        std::string orig(olen, '\0');
        rd(fi, &orig[0], olen);
        std::vector<Edit> raw, opt;
        for (uint32_t i = 0; i < n_ops; ++i) {
            uint8_t t, b; uint32_t num;
            rd(fi, &t, 1); rd(fi, &b, 1); rd(fi, &num, 1);
            if (t == 0) raw.push_back(Edit(SAME, (size_t)num));
            else if (t == 1) raw.push_back(Edit(INSERT, (size_t)(unsigned char)b));     // the constructor takes size_t; ins/del alias its low byte
            else raw.push_back(Edit(DELETE, (size_t)(unsigned char)b));
        }
        const uint64_t dis = Edit::optimizeEditScript(raw, opt);
        wr(fo, &dis, 1);
        const uint32_t n_new = (uint32_t)opt.size();
        wr(fo, &n_new, 1);
        for (const Edit &e : opt) {
            const uint8_t t = (uint8_t)e.editType, b = e.editType == SAME ? 0 : (uint8_t)e.editInfo.ins;
            const uint32_t num = e.editType == SAME ? (uint32_t)e.editInfo.num : 0;
            wr(fo, &t, 1); wr(fo, &b, 1); wr(fo, &num, 1);
        }
        for (const std::vector<Edit> *scr : {&raw, &opt}) {
            std::string res;
            Edits::applyEdits(orig.begin(), *scr, std::back_inserter(res));
            const uint32_t len = (uint32_t)res.size();
            wr(fo, &len, 1);
            wr(fo, res.data(), len);
        }
    }
    fclose(fi); fclose(fo);
    return 0;
}
