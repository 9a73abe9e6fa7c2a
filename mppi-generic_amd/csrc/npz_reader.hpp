/**
 * npz_reader.hpp — minimal reader for numpy .npz archives (host only), the on-disk format of the reference's models.
 *
 * The reference reads its network weights and costmaps with cnpy (an un-vendored submodule):
 *   FNNHelper::loadParams        include/mppi/utils/nn_helpers/fnn_helper.cu:96-174     keys dynamics_W{i}, dynamics_b{i} (float64)
 *   LSTMHelper::loadParams       include/mppi/utils/nn_helpers/lstm_helper.cu:514-585   keys [model/]{prefix}lstm/weight_{ih,hh}_l0, ...
 *   ARStandardCost::loadTrackData include/mppi/cost_functions/autorally/ar_standard_cost.cu:84-142  keys xBounds, yBounds,
 *                                                                                        pixelsPerMeter, channel0..3 (float32)
 * A .npz is a ZIP archive of .npy members (stored by numpy.savez, deflated by numpy.savez_compressed).  This reader
 * walks the ZIP central directory, inflates with zlib when needed, parses the NPY v1/v2/v3 header (descr, fortran_order,
 * shape) and converts little-endian f4 / f8 / i4 / i8 / u1 payloads to double.  No pickled objects, no ZIP64.
 */
#ifndef MPPI_AMD_NPZ_READER_HPP_
#define MPPI_AMD_NPZ_READER_HPP_

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace mppi
{
namespace npz
{
struct Array
{
  std::vector<int> shape;
  std::vector<double> data;  ///< C order
  std::string descr;
  size_t size() const
  {
    return data.size();
  }
};

inline uint16_t rd16(const unsigned char* p)
{
  return (uint16_t)(p[0] | (p[1] << 8));
}
inline uint32_t rd32(const unsigned char* p)
{
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/** parses one .npy image; returns false with `err` set on anything unexpected */
inline bool parseNpy(const std::vector<unsigned char>& b, Array& out, std::string& err)
{
  if (b.size() < 10 || memcmp(b.data(), "\x93NUMPY", 6) != 0)
  {
    err = "not an NPY member";
    return false;
  }
  const int major = b[6];
  size_t hlen, hoff;
  if (major == 1)
  {
    hlen = rd16(&b[8]);
    hoff = 10;
  }
  else
  {
    if (b.size() < 12)
    {
      err = "truncated NPY header";
      return false;
    }
    hlen = rd32(&b[8]);
    hoff = 12;
  }
  if (hoff + hlen > b.size())
  {
    err = "truncated NPY header";
    return false;
  }
  const std::string h((const char*)&b[hoff], hlen);
  auto field = [&](const char* key) -> std::string {
    const size_t k = h.find(key);
    if (k == std::string::npos)
      return "";
    size_t c = h.find(':', k);
    if (c == std::string::npos)
      return "";
    c++;
    while (c < h.size() && h[c] == ' ')
      c++;
    size_t e = c;
    if (c >= h.size())
      return "";
    if (h[c] == '(')
    {
      e = h.find(')', c);
      if (e == std::string::npos)
        return "";
      e++;
    }
    else if (h[c] == '\'')
    {
      e = h.find('\'', c + 1);
      if (e == std::string::npos)
        return "";
      e++;
    }
    else
      while (e < h.size() && h[e] != ',' && h[e] != '}')
        e++;
    return h.substr(c, e - c);
  };
  std::string descr = field("'descr'");
  if (descr.size() >= 2 && descr.front() == '\'')
    descr = descr.substr(1, descr.size() - 2);
  const std::string fortran = field("'fortran_order'");
  const std::string shape = field("'shape'");
  if (descr.empty() || shape.empty())
  {
    err = "NPY header without descr/shape";
    return false;
  }
  if (descr == "|O" || descr.find('O') != std::string::npos)
  {
    err = "pickled object arrays are not supported";
    return false;
  }
  out.descr = descr;
  out.shape.clear();
  size_t n = 1;
  for (size_t i = 0; i < shape.size();)
  {
    if (shape[i] >= '0' && shape[i] <= '9')
    {
      unsigned long long v = 0;
      while (i < shape.size() && shape[i] >= '0' && shape[i] <= '9')
      {
        v = v * 10 + (unsigned long long)(shape[i++] - '0');
        if (v > 0x7fffffffull)  // dimensions are ints downstream; also keeps the element count below from overflowing
        {
          err = "NPY shape dimension out of range";
          return false;
        }
      }
      if (v != 0 && n > (size_t)0x7fffffffffffull / (size_t)v)
      {
        err = "NPY element count out of range";
        return false;
      }
      out.shape.push_back((int)v);
      n *= (size_t)v;
    }
    else
      i++;
  }
  const char kind = descr.size() >= 2 ? descr[descr.size() - 2] : '?';
  const int width = descr.empty() ? 0 : descr.back() - '0';
  if (descr[0] == '>')
  {
    err = "big-endian arrays are not supported";
    return false;
  }
  const unsigned char* p = &b[hoff + hlen];
  if (width <= 0 || width > 8 || n > (b.size() - (hoff + hlen)) / (size_t)width)
  {
    err = "truncated NPY payload";
    return false;
  }
  std::vector<double> flat(n);
  for (size_t i = 0; i < n; i++)
  {
    const unsigned char* q = p + i * width;
    if (kind == 'f' && width == 8)
    {
      double v;
      memcpy(&v, q, 8);
      flat[i] = v;
    }
    else if (kind == 'f' && width == 4)
    {
      float v;
      memcpy(&v, q, 4);
      flat[i] = v;
    }
    else if (kind == 'i' && width == 8)
    {
      int64_t v;
      memcpy(&v, q, 8);
      flat[i] = (double)v;
    }
    else if (kind == 'i' && width == 4)
    {
      int32_t v;
      memcpy(&v, q, 4);
      flat[i] = (double)v;
    }
    else if ((kind == 'u' || kind == 'b') && width == 1)
    {
      flat[i] = (double)q[0];
    }
    else
    {
      err = "unsupported dtype " + descr;
      return false;
    }
  }
  if (fortran.find("True") != std::string::npos && out.shape.size() == 2)
  {  // column-major 2-D -> C order
    const int r = out.shape[0], c = out.shape[1];
    out.data.resize(n);
    for (int i = 0; i < r; i++)
      for (int j = 0; j < c; j++)
        out.data[(size_t)i * c + j] = flat[(size_t)j * r + i];
  }
  else
  {
    out.data.swap(flat);
  }
  return true;
}

/** loads every member of the archive; member "name.npy" is stored under key "name" */
inline bool load(const std::string& path, std::map<std::string, Array>& out, std::string& err)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f)
  {
    err = "cannot open " + path;
    return false;
  }
  fseek(f, 0, SEEK_END);
  const long fsize = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> buf(fsize > 0 ? (size_t)fsize : 0);
  const size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
  fclose(f);
  if (got != buf.size() || buf.size() < 22)
  {
    err = path + ": not a ZIP archive (a git-LFS pointer file?)";
    return false;
  }
  // end-of-central-directory record: signature 0x06054b50 within the last 64 KiB
  long eocd = -1;
  for (long i = (long)buf.size() - 22; i >= 0 && i >= (long)buf.size() - 65557; i--)
    if (rd32(&buf[i]) == 0x06054b50u)
    {
      eocd = i;
      break;
    }
  if (eocd < 0)
  {
    err = path + ": not a ZIP archive (a git-LFS pointer file?)";
    return false;
  }
  const int entries = rd16(&buf[eocd + 10]);
  size_t cd = rd32(&buf[eocd + 16]);
  for (int e = 0; e < entries; e++)
  {
    if (cd + 46 > buf.size() || rd32(&buf[cd]) != 0x02014b50u)
    {
      err = path + ": corrupt ZIP central directory";
      return false;
    }
    const int method = rd16(&buf[cd + 10]);
    const size_t csize = rd32(&buf[cd + 20]), usize = rd32(&buf[cd + 24]);
    const int nlen = rd16(&buf[cd + 28]), xlen = rd16(&buf[cd + 30]), clen = rd16(&buf[cd + 32]);
    const size_t lho = rd32(&buf[cd + 42]);
    if (cd + 46 + (size_t)nlen + (size_t)xlen + (size_t)clen > buf.size())
    {
      err = path + ": corrupt ZIP central directory (entry runs past the end of the file)";
      return false;
    }
    std::string name((const char*)&buf[cd + 46], nlen);
    cd += 46 + nlen + xlen + clen;
    if (csize == 0xffffffffu || usize == 0xffffffffu)
    {
      err = path + ": ZIP64 members are not supported";
      return false;
    }
    if (lho + 30 > buf.size() || rd32(&buf[lho]) != 0x04034b50u)
    {
      err = path + ": corrupt ZIP local header";
      return false;
    }
    const size_t data_off = lho + 30 + rd16(&buf[lho + 26]) + rd16(&buf[lho + 28]);
    // deflate expands at most ~1032x; an uncompressed size beyond that (or beyond 1 GiB) is a corrupt / hostile directory
    if (data_off > buf.size() || csize > buf.size() - data_off || usize > ((size_t)1 << 30) ||
        (method == 8 && usize > 1040 * csize + 64))
    {
      err = path + ": truncated ZIP member";
      return false;
    }
    std::vector<unsigned char> raw;
    if (method == 0)
    {
      raw.assign(buf.begin() + data_off, buf.begin() + data_off + csize);
    }
    else if (method == 8)
    {
      raw.resize(usize);
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK)
      {
        err = "zlib inflateInit2 failed";
        return false;
      }
      zs.next_in = &buf[data_off];
      zs.avail_in = (uInt)csize;
      zs.next_out = raw.data();
      zs.avail_out = (uInt)usize;
      const int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END)
      {
        err = path + ": inflate failed for member " + name;
        return false;
      }
    }
    else
    {
      err = path + ": unsupported ZIP compression method for member " + name;
      return false;
    }
    if (name.size() > 4 && name.substr(name.size() - 4) == ".npy")
      name.resize(name.size() - 4);
    Array a;
    if (!parseNpy(raw, a, err))
    {
      err = path + " [" + name + "]: " + err;
      return false;
    }
    out[name] = std::move(a);
  }
  return true;
}
}  // namespace npz
}  // namespace mppi
#endif
