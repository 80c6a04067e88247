// Params::use_silver_screen: RGBToYUV420 of the reference (guetzli/preprocess_downsample.cc:
// 283-476) -- the YUV 4:2:0 samples whose decoded image has the luma of the original when
// averaged in LINEAR light, found by 20 fixed-point iterations through the decoder model.
// Host code on purpose: the iteration is made of std::pow calls on arbitrary float arguments,
// and bit-identical results need the same libm as the reference uses (glibc's pow); it runs
// once per image, row-parallel on the worker pool.  Its three output planes go to the device
// (gz_set_orig_from_planes_420), where SetDownsampledCoefficients makes coefficients of them.
#pragma once
#include <stdint.h>

#include <vector>

namespace guetzli_amd {

// rgb: packed 8-bit sRGB (w*h*3) = OutputImage::ToSRGB() of the unquantised image.
// y, u, v: w*h floats each (u and v already box-upsampled to full resolution, as
// RGBToYUV420 returns them).
void SilverScreenYUV420(const uint8_t* rgb, int w, int h, std::vector<float>* y,
                        std::vector<float>* u, std::vector<float>* v);

}  // namespace guetzli_amd
