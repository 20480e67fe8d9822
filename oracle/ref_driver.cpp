// ref_driver.cpp -- thin extern "C" driver around the REFERENCE arithmetic coder.
// TEST INFRASTRUCTURE ONLY.  It is linked against the reference's own
// extension/ArithmeticCoder.cpp and extension/BitIoStream.cpp, compiled from where they
// lie under /root/reference (see oracle/Makefile target `ref`); outputs go to oracle/_ref/
// (git-ignored).  No reference source is copied into this repository.
//
// The loops below restate what Coder::my_encoder_slice[_mask] / my_decoder_slice[_mask]
// do (extension/coder.cpp:30-113) without the at::Tensor glue; coder.cpp itself needs
// torch + CUDA headers and is treated as unbuildable here.
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>
#include "ArithmeticCoder.h"
#include "BitIoStream.h"

extern "C" {

// Encode `nslices` consecutive slices into one stream.  tables: int32 [total_syms][ncode+1],
// labels int32 [total_syms], mask float [total_syms] or NULL.  Returns number of bytes written
// (or -1 if cap too small, -2 on a coder exception).
long ref_ac_encode(const int *tables, int ncode, const int *labels, const float *mask, long num,
                   unsigned char *out, long cap) {
    try {
        std::ostringstream os(std::ios::binary);
        BitOutputStream bout(os);
        ArithmeticEncoder enc(32, bout);
        std::vector<uint32_t> t(ncode + 1);
        for (long i = 0; i < num; ++i) {
            if (mask && mask[i] < 0.5f) continue;
            for (int j = 0; j <= ncode; ++j) t[j] = static_cast<uint32_t>(tables[i * (ncode + 1) + j]);
            enc.write(t.data(), ncode, t[ncode], static_cast<uint32_t>(labels[i]));
        }
        enc.finish();
        bout.finish();
        std::string s = os.str();
        if ((long)s.size() > cap) return -1;
        std::memcpy(out, s.data(), s.size());
        return (long)s.size();
    } catch (const char *) {
        return -2;
    }
}

// Decode `num` symbols (masked slots receive file_value).  Returns 0, or -2 on a coder exception.
int ref_ac_decode(const unsigned char *bytes, long n, const int *tables, int ncode, const float *mask,
                  float file_value, long num, float *out) {
    try {
        std::istringstream is(std::string(reinterpret_cast<const char *>(bytes), (size_t)n), std::ios::binary);
        BitInputStream bin(is);
        ArithmeticDecoder dec(32, bin);
        std::vector<uint32_t> t(ncode + 1);
        for (long i = 0; i < num; ++i) {
            if (mask && mask[i] < 0.5f) { out[i] = file_value; continue; }
            for (int j = 0; j <= ncode; ++j) t[j] = static_cast<uint32_t>(tables[i * (ncode + 1) + j]);
            out[i] = static_cast<float>(dec.read(t.data(), ncode, t[ncode]));
        }
        return 0;
    } catch (const char *) {
        return -2;
    }
}
}
