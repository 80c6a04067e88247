#include "silver_screen.h"

#include <math.h>

#include <algorithm>

#include "parallel.h"

namespace guetzli_amd {

namespace {

// preprocess_downsample.cc:283-319 -- every expression with the reference's operand types
inline float Clip(float v) { return std::max(0.0f, std::min(255.0f, v)); }
inline float ToY(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }
inline float ToU(float r, float g, float b) { return -0.16874f * r - 0.33126f * g + 0.5f * b + 128.0f; }
inline float ToV(float r, float g, float b) { return 0.5f * r - 0.41869f * g - 0.08131f * b + 128.0f; }
inline float GammaToLinear(float x) { return static_cast<float>(pow(x / 255.0f, 2.2)); }
inline float LinearToGamma(float x) { return static_cast<float>(255.0 * pow(x, 1.0 / 2.2)); }

// Rows [0, n) in chunks on the worker pool (every stage below is independent per output row).
template <class F>
void Rows(int n, const F& f) {
  WorkerPool& pool = WorkerPool::Get();
  const int chunks = std::max(1, std::min(n, 4 * pool.size()));
  const int per = (n + chunks - 1) / chunks;
  pool.Run(chunks, [&](int c) {
    const int r0 = c * per, r1 = std::min(n, r0 + per);
    for (int r = r0; r < r1; ++r) f(r);
  });
}

struct Planes {
  int w, h, w2, h2;
  std::vector<float> rgb;   // packed, w*h*3
};

// LinearlyAveragedLuma, :321-330
void Luma(const std::vector<float>& rgb, int w, int h, std::vector<float>* y) {
  y->resize((size_t)w * h);
  Rows(h, [&](int r) {
    for (int x = 0; x < w; ++x) {
      const size_t i = (size_t)r * w + x;
      (*y)[i] = LinearToGamma(ToY(GammaToLinear(rgb[3 * i]), GammaToLinear(rgb[3 * i + 1]),
                                  GammaToLinear(rgb[3 * i + 2])));
    }
  });
}

// RGBToYUV(LinearlyDownsample2x2(rgb)), :332-367
void DownsampledYUV(const std::vector<float>& rgb, int w, int h, std::vector<float> yuv[3]) {
  const int w2 = (w + 1) / 2, h2 = (h + 1) / 2;
  for (int c = 0; c < 3; ++c) yuv[c].resize((size_t)w2 * h2);
  Rows(h2, [&](int y) {
    for (int x = 0; x < w2; ++x) {
      float px[3];
      for (int i = 0; i < 3; ++i) {
        float acc = 0.0;
        for (int iy = 0; iy < 2; ++iy)
          for (int ix = 0; ix < 2; ++ix) {
            const int yy = std::min(h - 1, 2 * y + iy), xx = std::min(w - 1, 2 * x + ix);
            acc += GammaToLinear(rgb[3 * ((size_t)yy * w + xx) + i]);
          }
        px[i] = LinearToGamma(0.25f * acc);
      }
      const size_t o = (size_t)y * w2 + x;
      yuv[0][o] = ToY(px[0], px[1], px[2]);
      yuv[1][o] = ToU(px[0], px[1], px[2]);
      yuv[2][o] = ToV(px[0], px[1], px[2]);
    }
  });
}

// Upsample2x2 (box), :384-402: out(yy, xx) = in(yy / 2, xx / 2)
void BoxUpsample(const std::vector<float>& in, int w, int h, std::vector<float>* out) {
  const int w2 = (w + 1) / 2;
  out->resize((size_t)w * h);
  Rows(h, [&](int y) {
    for (int x = 0; x < w; ++x) (*out)[(size_t)y * w + x] = in[(size_t)(y / 2) * w2 + x / 2];
  });
}

// Blur ("fancy upsample" on the box-upsampled plane), :405-426
void Fancy(const std::vector<float>& img, int w, int h, std::vector<float>* out) {
  out->resize((size_t)w * h);
  Rows((h + 1) / 2, [&](int cy) {
    const int y0 = 2 * cy;
    for (int x0 = 0; x0 < w; x0 += 2)
      for (int iy = 0; iy < 2 && y0 + iy < h; ++iy)
        for (int ix = 0; ix < 2 && x0 + ix < w; ++ix) {
          const int dy = 4 * iy - 2, dx = 4 * ix - 2;
          const int x1 = std::min(w - 1, std::max(0, x0 + dx));
          const int y1 = std::min(h - 1, std::max(0, y0 + dy));
          (*out)[(size_t)(y0 + iy) * w + x0 + ix] =
              (9.0f * img[(size_t)y0 * w + x0] + 3.0f * img[(size_t)y0 * w + x1] +
               3.0f * img[(size_t)y1 * w + x0] + 1.0f * img[(size_t)y1 * w + x1]) / 16.0f;
        }
  });
}

}  // namespace

// RGBToYUV420, :452-476
void SilverScreenYUV420(const uint8_t* rgb_in, int w, int h, std::vector<float>* gy,
                        std::vector<float>* gu, std::vector<float>* gv) {
  const size_t n = (size_t)w * h;
  std::vector<float> rgbf(3 * n);
  for (size_t i = 0; i < 3 * n; ++i) rgbf[i] = static_cast<float>(rgb_in[i]);
  std::vector<float> y_target, yuv_target[3];
  Luma(rgbf, w, h, &y_target);
  DownsampledYUV(rgbf, w, h, yuv_target);
  std::vector<float> guess_y, guess_u = yuv_target[1], guess_v = yuv_target[2];
  BoxUpsample(yuv_target[0], w, h, &guess_y);
  std::vector<float> up_u, up_v, fu, fv, rec(3 * n), y_rec, yuv_rec[3];
  for (int iter = 0; iter < 20; ++iter) {
    // YUV420ToRGB, :428-437
    BoxUpsample(guess_u, w, h, &up_u);
    BoxUpsample(guess_v, w, h, &up_v);
    Fancy(up_u, w, h, &fu);
    Fancy(up_v, w, h, &fv);
    Rows(h, [&](int r) {
      for (int x = 0; x < w; ++x) {
        const size_t i = (size_t)r * w + x;
        const float y = guess_y[i], u = fu[i], v = fv[i];
        rec[3 * i] = Clip(y + 1.402f * (v - 128.0f));
        rec[3 * i + 1] = Clip(y - 0.344136f * (u - 128.0f) - 0.714136f * (v - 128.0f));
        rec[3 * i + 2] = Clip(y + 1.772f * (u - 128.0f));
      }
    });
    Luma(rec, w, h, &y_rec);
    DownsampledYUV(rec, w, h, yuv_rec);
    // UpdateGuess, :439-448
    Rows(h, [&](int r) {
      for (int x = 0; x < w; ++x) {
        const size_t i = (size_t)r * w + x;
        guess_y[i] = Clip(guess_y[i] - (y_rec[i] - y_target[i]));
      }
    });
    const size_t n2 = guess_u.size();
    for (size_t i = 0; i < n2; ++i) {
      guess_u[i] = Clip(guess_u[i] - (yuv_rec[1][i] - yuv_target[1][i]));
      guess_v[i] = Clip(guess_v[i] - (yuv_rec[2][i] - yuv_target[2][i]));
    }
  }
  *gy = guess_y;
  BoxUpsample(guess_u, w, h, gu);
  BoxUpsample(guess_v, w, h, gv);
}

}  // namespace guetzli_amd
