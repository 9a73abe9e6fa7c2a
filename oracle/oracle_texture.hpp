/**
 * oracle_texture.hpp — CPU restatement of the reference's 2-D texture helper lookups.  TEST INFRASTRUCTURE ONLY.
 * Follows the reference's HOST path, which is itself an fp32 restatement of the texture unit:
 *   TextureHelper::worldPoseToMapPose / mapPoseToTexCoord   utils/texture_helpers/texture_helper.cu:94-124
 *   TwoDTextureHelper::queryTextureCPU                       utils/texture_helpers/two_d_texture_helper.cu:151-245
 * pinned on the known answers of tests/texture_helpers/two_d_texture_helper_test.cu:368-541 (tests/test_texture_helper.py).
 */
#ifndef MPPI_ORACLE_TEXTURE_HPP_
#define MPPI_ORACLE_TEXTURE_HPP_

#include <algorithm>
#include <cmath>
#include <vector>

namespace oracle
{
struct Texture2D
{
  const float* values = nullptr;  // [height][width][channels]
  int width = 0, height = 0, channels = 1;
  int address_mode[2] = { 0, 0 };  // 0 clamp, 1 border
  int filter_mode = 0;             // 0 linear, 1 point
  float border_color[4] = { 0, 0, 0, 0 };
  float origin[3] = { 0, 0, 0 };
  float rot[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
  float resolution[3] = { 1, 1, 1 };

  void worldToMap(const float* in, float* out) const
  {
    const float diff[3] = { in[0] - origin[0], in[1] - origin[1], in[2] - origin[2] };
    for (int r = 0; r < 3; r++)
      out[r] = (rot[r][0] * diff[0] + rot[r][1] * diff[1] + rot[r][2] * diff[2]);
  }
  void mapToTex(const float* in, float* out) const
  {
    out[0] = in[0] / resolution[0];
    out[1] = in[1] / resolution[1];
    out[2] = in[2] / resolution[2];
    out[0] /= (float)width;
    out[1] /= (float)height;
  }
  void query(const float* point, float* result) const
  {
    float qx = point[0] * (float)width, qy = point[1] * (float)height;
    qx = qx - 0.5f;
    qy = qy - 0.5f;
    // the reference has no NaN rule (a NaN reaches an int conversion); the product maps it to texel 0, and so does this
    qx = (qx == qx) ? qx : 0.0f;
    qy = (qy == qy) ? qy : 0.0f;
    bool outside = false;
    if (address_mode[0] == 0)
    {
      if (qx > (float)(width - 1))
        qx = (float)(width - 1);
      else if (qx <= 0.0f)
        qx = 0.0f;
    }
    else if (qx > (float)(width - 1) || qx <= 0.0f)
      outside = true;
    if (address_mode[1] == 0)
    {
      if (qy > (float)(height - 1))
        qy = (float)(height - 1);
      else if (qy <= 0.0f)
        qy = 0.0f;
    }
    else if (qy > (float)(height - 1) || qy <= 0.0f)
      outside = true;
    for (int ch = 0; ch < channels; ch++)
    {
      if (outside)
      {
        result[ch] = border_color[ch];
        continue;
      }
      auto at = [&](int row, int col) { return values[((size_t)row * width + col) * channels + ch]; };
      if (filter_mode == 1)
      {
        result[ch] = at((int)std::round(qy), (int)std::round(qx));
        continue;
      }
      // 1-texel axes (not covered by the reference, which would index texel -1): both taps read texel 0
      const int x_min = std::max(std::min((int)std::floor(qx), width - 2), 0), x_max = std::min(x_min + 1, width - 1);
      const int y_min = std::max(std::min((int)std::floor(qy), height - 2), 0), y_max = std::min(y_min + 1, height - 1);
      const float wx0 = x_max > x_min ? (x_max - qx) / (x_max - x_min) : 1.0f, wx1 = x_max > x_min ? (qx - x_min) / (x_max - x_min) : 0.0f;
      const float wy0 = y_max > y_min ? (y_max - qy) / (y_max - y_min) : 1.0f, wy1 = y_max > y_min ? (qy - y_min) / (y_max - y_min) : 0.0f;
      const float y_min_interp = at(y_min, x_min) * wx0 + at(y_min, x_max) * wx1;
      const float y_max_interp = at(y_max, x_min) * wx0 + at(y_max, x_max) * wx1;
      result[ch] = y_min_interp * wy0 + y_max_interp * wy1;
    }
  }
};
}  // namespace oracle

#endif
