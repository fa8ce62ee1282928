// ASan/UBSan driver for the host-only I/O code: writers + CSV reader round trip on hostile inputs.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../seekr_amd/csrc/common.hpp"
int main() {
    std::mt19937_64 g(1);
    std::vector<float> a(300 * 257);
    for (auto& x : a) {
        uint32_t b = (uint32_t)g();
        memcpy(&x, &b, 4);
    }
    std::vector<double> d(a.begin(), a.end());
    for (int mode = 0; mode < 3; mode++) {
        if (skr_host_save_csv(a.data(), SKR_F32, 300, 257, mode, 3, "/tmp/seekr_san/f.csv")) return 1;
        if (skr_host_save_csv(d.data(), SKR_F64, 300, 257, mode, 3, "/tmp/seekr_san/d.csv")) return 2;
    }
    std::string rl, cl;
    for (int i = 0; i < 300; i++) rl += (i ? "\n" : "") + std::string(">r\"") + std::to_string(i) + ",x";
    for (int i = 0; i < 257; i++) cl += (i ? "\n" : "") + std::string("c") + std::to_string(i);
    if (skr_host_save_csv_labelled(a.data(), SKR_F32, 300, 257, rl.c_str(), cl.c_str(), 4, "/tmp/seekr_san/l.csv")) return 3;
    if (skr_host_save_npy(a.data(), SKR_F32, 300, 257, 0, "/tmp/seekr_san/a.npy")) return 4;
    skr_csv* csv = nullptr;
    int rc = skr_csv_read("/tmp/seekr_san/l.csv", 4, &csv);
    printf("read of raw-bit-pattern file: rc=%d (%s)\n", rc, rc ? skr_last_error() : "ok");
    if (csv) skr_csv_free(csv);
    // a well-formed file
    for (auto& x : a) x = (float)((int)(g() % 2000000) - 1000000) / 1024.0f;
    if (skr_host_save_csv_labelled(a.data(), SKR_F32, 300, 257, rl.c_str(), cl.c_str(), 4, "/tmp/seekr_san/l.csv")) return 5;
    rc = skr_csv_read("/tmp/seekr_san/l.csv", 4, &csv);
    printf("read of clean file: rc=%d\n", rc);
    if (rc) return 6;
    int64_t r, c, need;
    skr_csv_shape(csv, &r, &c);
    std::vector<double> v(r * c);
    skr_csv_values(csv, v.data());
    for (size_t i = 0; i < v.size(); i++)
        if ((float)v[i] != a[i]) { printf("mismatch at %zu\n", i); return 7; }
    skr_csv_labels(csv, 0, nullptr, 0, &need);
    std::vector<char> buf(need);
    skr_csv_labels(csv, 0, buf.data(), need, &need);
    skr_csv_free(csv);
    // hostile files
    const char* bad[] = {"", ",a\n", ",a,b\n\"x,1,2\n", ",a\n>r,1e999999999\n", ",a\n>r,--1\n", ",a\n>r,1,2,3\n", "\n\n\n",
                         ",a\n\"q\"\"\",5\n", ",a\r\n>r,\r\n", ",a\n>r,1.", ",\"a\n"};
    for (const char* t : bad) {
        FILE* fh = fopen("/tmp/seekr_san/b.csv", "wb"); fwrite(t, 1, strlen(t), fh); fclose(fh);
        csv = nullptr;
        rc = skr_csv_read("/tmp/seekr_san/b.csv", 2, &csv);
        if (csv) skr_csv_free(csv);
    }
    // ---- FASTA reader: hostile and random files, parsed in forced multi-piece mode.  Without a GPU the call
    // ends in an error at the upload, after the parser, the stitcher and the packer have all run.
    skr_ctx* fake = new skr_ctx();
    const char* fastas[] = {"", ">", ">h", ">h\n", "ACGT", ">a\nAC\n>b\n", ">a\r\nAC\r\n\r\n>b\r\nGT", ">a\rACGT\r>b\rTT\r",
                            ">a\n>b\nAC\n", "\n>a\nAC", ">a\n \t \nAC\n", ">a\nACGT\n>b\nacgtnnnn\n>c\nA\n"};
    setenv("SEEKR_FASTA_PIECE_BYTES", "3", 1);
    int parsed = 0;
    auto try_file = [&](const std::string& text) {
        FILE* fh = fopen("/tmp/seekr_san/f.fa", "wb");
        fwrite(text.data(), 1, text.size(), fh);
        fclose(fh);
        skr_seqs* sq = nullptr;
        const int rc2 = skr_seqs_from_fasta(fake, "/tmp/seekr_san/f.fa", "AGTC", &sq);
        if (rc2 == SKR_OK && sq) skr_seqs_free(sq);
        parsed++;
    };
    for (const char* t : fastas) try_file(t);
    const char alphabet[] = "ACGTNacgt>\n\r \t;|0";
    for (int rep = 0; rep < 3000; rep++) {
        std::string text;
        const int len = (int)(g() % 400);
        if (g() % 4) text += ">first\n";
        for (int i = 0; i < len; i++) {
            const uint64_t r = g();
            text += (r % 13 == 0) ? '\n' : (r % 47 == 0 ? '>' : alphabet[r % (sizeof(alphabet) - 1)]);
        }
        if (rep % 3 == 0) setenv("SEEKR_FASTA_PIECE_BYTES", std::to_string(1 + g() % 64).c_str(), 1);
        try_file(text);
    }
    printf("fasta reader: %d files parsed under the sanitizers\n", parsed);
    puts("sanitizer driver done");
    return 0;
}
