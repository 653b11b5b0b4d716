// JPEG and PNG decoding for texture files (host code; see image_decode.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

namespace rt {

struct Image {
    int w = 0, h = 0;
    int channels = 0;         // as stored in the file: 1 grey, 2 grey + alpha, 3 RGB, 4 RGBA
    std::vector<uint8_t> px;  // tightly packed, row 0 = top row of the picture
};

// Decodes a Huffman-coded 8-bit JPEG (sequential or progressive; 1, 3 or 4 components) or a PNG (any colour type and bit
// depth, interlaced or not) held in memory.
// The bytes equal what stbi_load(..., req_comp = 0) of stb_image v2.27 returns for the same file.
bool decodeImage(const uint8_t* data, size_t n, Image* out, std::string* err);
// Three channels per texel: grey replicated, alpha dropped.
void toRgb(const Image& in, std::vector<uint8_t>* rgb);

}  // namespace rt
