// Baseline-sequential JPEG serialisation used by the host search driver: what
// OutputImage::SaveToJpegData + WriteJpeg produce in the reference
// (output_image.cc:348-409, jpeg_data_writer.cc:33-553, entropy_encode.cc:25-145,
// jpeg_bit_writer.h:31-108, jpeg_data.cc:71-102).  Byte-exact by contract: the size of
// every candidate feeds ScoreJPEG and the bytes of the winner are the product's output.
//
// Host-side C++ (serial entropy coding); written from scratch around a
// flat coefficient layout (the device layout of include/guetzli_amd.h), not around the
// reference's JPEGData object model.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace guetzli_amd {

constexpr int kBlock = 64;
constexpr int kHistoSize = 257;   // 256 symbols + the reserved all-ones code

extern const int kNaturalOrder[64];   // zig-zag position -> natural index
extern const int kZigZagOrder[64];    // natural index -> zig-zag position

// Symbol statistics with every real symbol counted twice and one reserved symbol counted
// once, so that the reserved symbol always ends up with the longest (all-ones) code
// (JpegHistogram, jpeg_data_writer.h:53-85).
struct SymbolHistogram {
  uint32_t counts[kHistoSize];
  SymbolHistogram() { Clear(); }
  void Clear();
  void Add(int symbol, int weight = 1) { counts[symbol] += 2 * weight; }
  void Merge(const SymbolHistogram& other);
  int NumSymbols() const;
};

// Length-limited Huffman depths (CreateHuffmanTree, entropy_encode.cc:73-145).
// stream: which sequence of slowly changing histograms this one belongs to (0..7; the construction
// starts from the last order of the stream's symbols), -1 for none.  The result does not depend on it.
void HuffmanDepths(const uint32_t* counts, size_t length, int tree_limit, uint8_t* depth, int stream = -1);

size_t HistogramHeaderBits(const SymbolHistogram& h);                       // :218-226
size_t HistogramEntropyBits(const SymbolHistogram& h, const uint8_t* depth);  // :228-239
// Greedy pairwise clustering from the back (ClusterHistograms, :295-342).  Returns the
// estimated size in bytes.
size_t ClusterHistograms(SymbolHistogram* histo, size_t* num, int* histo_indexes,
                         uint8_t* depth);

// Entropy-size model of the AC coefficients (processor.cc:497-525): the histograms clustered,
// depths[i * kHistoSize ..] = the code lengths histogram i is coded with; returns the header
// bytes of the clustered codes.  EntropyDataSize: the data bytes under those code lengths.
size_t EntropyCodes(const SymbolHistogram* histo, int n, uint8_t* depths /* n * kHistoSize */);
size_t EntropyDataSize(const SymbolHistogram* histo, int n, const uint8_t* depths);

// One AC block's symbols (UpdateACHistogramForDCTBlock :197-216 / processor.cc:471-495):
// coefficients are DEQUANTISED values, q the component's quant matrix (null = already
// quantised).
// With depth / raw_bits: *raw_bits follows HistogramRawBits(h, depth) through the update (the
// search's size estimate after every coefficient step without a pass over the histograms).
void AddBlockACSymbols(const int16_t* block, const int* q, int weight, SymbolHistogram* h,
                       const uint8_t* depth = nullptr, int64_t* raw_bits = nullptr);
// HistogramEntropyBits == EntropyBitsFromRaw(HistogramRawBits(h, depth)).
// AddBlockACSymbols(blk, q, -1) + blk[k] = newval + AddBlockACSymbols(blk, q, +1) without the two
// passes (k >= 1; blk holds the old value and is not written).
void ReplaceCoeffACSymbols(const int16_t* blk, const int* q, int k, int newval, SymbolHistogram* h,
                           const uint8_t* depth, int64_t* raw_bits);
// The same symbol changes as a list instead of an update: changes[i] = +(symbol + 1) for an occurrence
// that enters the block's scan, -(symbol + 1) for one that leaves it; at most kMaxCoeffACSymbolChanges
// entries (two windows of: up to three ZRL codes and a symbol for the coefficient, the same for its
// successor, or an end-of-block code).  Returns their number.  blk is not written.
const int kMaxCoeffACSymbolChanges = 16;
int CoeffACSymbolChanges(const int16_t* blk, const int* q, int k, int newval, int16_t* changes);
int64_t HistogramRawBits(const SymbolHistogram& h, const uint8_t* depth);
size_t EntropyBitsFromRaw(int64_t raw);

// A frame ready to be written: quantised coefficient planes + tables.
struct QuantTable {
  int values[64];
  int precision;   // 0: 8 bit, 1: 16 bit
  int index;
};
// What a JPEG INPUT contributes to every file written for it (JPEGData fields that
// WriteJpeg / JpegHeaderSize read, jpeg_data_writer.cc:52-72,269-293,540-553); null for RGB
// input, whose JPEGData carries only the fixed JFIF APP0 and component ids 0, 1, 2.
struct FrameMeta {
  bool strip = true;                    // Params::clear_metadata
  std::vector<std::string> app_data;    // marker byte + segment
  std::vector<std::string> com_data;    // segment with its length
  std::string tail_data;                // bytes after EOI
};
struct Frame {
  int width = 0, height = 0, bw = 0, bh = 0;   // bw x bh: the 8x8 (luma) grid of the image
  int ncomp = 3;
  // Sampling: component c has samp[c] x samp[c] blocks per MCU (4:4:4: 1,1,1 -- 4:2:0: 2,1,1),
  // its coefficient plane holds cw[c] x ch[c] blocks = whole MCUs (mcu_cols * samp[c] ...),
  // padding included (OutputImage::SaveToJpegData, output_image.cc:348-409).
  int samp[3] = {1, 1, 1};
  int mcu_cols = 0, mcu_rows = 0;
  int cw[3] = {0, 0, 0}, ch[3] = {0, 0, 0};
  int comp_id[3] = {0, 1, 2};           // SaveToJpegData numbers them (output_image.cc:376); the
                                        // input JPEG as read keeps its own
  const FrameMeta* meta = nullptr;
  std::vector<int16_t> coeffs[3];    // [nb][64], quantised values (coeff / quant)
  std::vector<QuantTable> quant;
  int quant_idx[3] = {0, 0, 0};
};

// OutputImage::SaveToJpegData + SaveQuantTables for a 4:4:4 image given by dequantised
// coefficients (device layout, [3][nb][64]) and its quant matrices.
void FrameFromImage(const int16_t* coeffs, const int q[3][64], int w, int h, Frame* f);
// The same for a frame with chroma subsampling factor `factor` (1: as above; 2: YUV 4:2:0,
// coefficients = nb luma blocks, nbc Cb, nbc Cr): luma padded to whole MCUs with the
// reference's padding blocks (all AC zero, DC = the DC of the block before in raster order).
void FrameFromImageFactor(const int16_t* coeffs, const int q[3][64], int w, int h, int factor,
                          Frame* f);
// The q=1 "original" JPEGData of EncodeRGBToJpeg (jpeg_data_encoder.cc:66-117): three
// separate all-ones tables that all carry table index 0.
void FrameFromOriginal(const int16_t* coeffs, int w, int h, Frame* f);

// The tables of a frame without its coefficients (enough for HeaderSize / BuildJpegHead):
// q = the image's quant matrices, or null for the q=1 "original" flavour.
void FrameTables(const int q[3][64], int w, int h, int ncomp, Frame* f);
void FrameTablesFactor(const int q[3][64], int w, int h, int ncomp, int factor, Frame* f);

void BuildDCHistograms(const Frame& f, SymbolHistogram* histo);   // :241-265
void BuildACHistograms(const Frame& f, SymbolHistogram* histo);   // :267-275
size_t HeaderSize(const Frame& f);                                // JpegHeaderSize :278-303
size_t EstimateDCSize(const Frame& f);                            // processor.cc:527-535

// Everything of the file in front of the entropy-coded scan -- SOI, APP0, DQT, SOF, DHT, SOS
// (jpeg_data_writer.cc:33-127,361-444) -- built from the frame's tables (f.coeffs is not
// read) and the symbol statistics, plus the per-component codes the scan is written with
// (depth 255 = symbol has no code).  The scan itself comes from the device
// (gz_jpeg_scan) or, in WriteJpeg, from the host bit writer.
struct JpegHead {
  int ncomp;
  std::string bytes;
  uint8_t depth[2][3][256];    // (DC, AC) x component x symbol
  uint16_t code[2][3][256];
};
bool BuildJpegHead(const Frame& f, const SymbolHistogram* dc_histo,
                   const SymbolHistogram* ac_histo, JpegHead* head);

// WriteJpeg (jpeg_data_writer.cc:540-553): a fixed JFIF APP0 unless the frame carries the
// metadata of a JPEG input that is to be kept, in which case its APPn / COM segments and its
// tail follow the reference's order.  Returns false on an internal inconsistency.
bool WriteJpeg(const Frame& f, std::string* out);

}  // namespace guetzli_amd
