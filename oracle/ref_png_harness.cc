// TEST INFRASTRUCTURE ONLY.  The reference's PNG front end behind a C shim: ReadPNG lives in
// an anonymous namespace of guetzli/guetzli.cc (:47-152), next to main(), so this translation
// unit includes that file as it lies under /root/reference (main renamed) and exports one
// function.  Built by oracle/Makefile into oracle/_ref/libgz_ref_png.so against the system's
// libpng16.so.16 (headers: /opt/conda/include/libpng16, copied to oracle/_ref/inc at build
// time).  Nothing under guetzli_amd/ may link, load or execute it.
#define main guetzli_reference_cli_main
#include "guetzli/guetzli.cc"
#undef main

#include <string.h>

extern "C" long ref_read_png(const unsigned char* data, long len, int* wh, unsigned char* out,
                             long cap) {
  std::string s((const char*)data, (size_t)len);
  std::vector<uint8_t> rgb;
  int w = 0, h = 0;
  if (!ReadPNG(s, &w, &h, &rgb)) return -1;
  wh[0] = w;
  wh[1] = h;
  if ((long)rgb.size() <= cap) memcpy(out, rgb.data(), rgb.size());
  return (long)rgb.size();
}
