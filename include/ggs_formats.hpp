// ggs_formats.hpp -- the on-disk formats of the reference's Java driver, written from C++, next to the
// C++ sampler mirror of ggs_sampler.hpp: a native run can leave the files the Java driver would have left without a
// copy-back through the JVM.  Header-only, C++17, no dependency on the GPU library.  Same rules as the Python writers of
// ldagroupedgibbssampler_amd/formats.py (tests/test_formats.py compares the two byte for byte):
//
//   byte-exact by construction
//     *_<rows>_<cols>_%05d.BINARY      util/LDAUtils.java:1120-1173  big-endian doubles / ints (ByteBuffer.putDouble/putInt);
//                                      the int writer maps the file at 8*rows*cols bytes and fills half of it (:1161-1173)
//     ASCII integer matrices           LDAUtils.java:1175-1197
//     z_<iteration>.csv                topics/UncollapsedParallelLDA.java:945-968
//   restated from the JDK's documented behaviour -- parity UNPINNED (no JVM here, no output file in the reference)
//     Double.toString                  log-likelihood.txt / test_held_out_log_likelihood.txt lines, LDAUtils.java:928-940,971-979
//     String.format("%.nf")            LDAUtils.formatDouble, LDAUtils.java:1199-1207; log-posterior.txt, :955-968
//     DecimalFormat("00.###E0")        magnitudes below 1e-4, LDAUtils.java:1200-1201,1225
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace ggs {
namespace formats {

// ---- binary matrices -------------------------------------------------------------------------------------------
inline std::string binary_matrix_name(const std::string &prefix, int64_t rows, int64_t cols, int iteration) {   // LDAUtils.java:1127-1129
  char tail[64];
  std::snprintf(tail, sizeof tail, "_%lld_%lld_%05d.BINARY", (long long)rows, (long long)cols, iteration);
  return prefix + tail;
}
inline void put_be(std::string &out, uint64_t bits, int bytes) {
  for (int b = bytes - 1; b >= 0; --b) out.push_back((char)((bits >> (8 * b)) & 0xff));
}
inline void write_file(const std::string &fn, const std::string &bytes) {
  std::ofstream f(fn, std::ios::binary | std::ios::trunc);
  if (!f) throw std::runtime_error("cannot open " + fn);
  f.write(bytes.data(), (std::streamsize)bytes.size());
}
inline void write_binary_double_matrix(const double *m, int64_t rows, int64_t cols, const std::string &fn) {   // LDAUtils.java:1132-1143
  std::string out;
  out.reserve((size_t)rows * cols * 8);
  for (int64_t i = 0; i < rows * cols; ++i) {
    uint64_t bits;
    std::memcpy(&bits, &m[i], 8);
    put_be(out, bits, 8);
  }
  write_file(fn, out);
}
inline void write_binary_int_matrix(const int32_t *m, int64_t rows, int64_t cols, const std::string &fn) {     // LDAUtils.java:1161-1173
  std::string out;
  out.reserve((size_t)rows * cols * 8);
  for (int64_t i = 0; i < rows * cols; ++i) put_be(out, (uint32_t)m[i], 4);
  out.append((size_t)rows * cols * 4, '\0');          // the file is mapped at the double writer's size: the second half stays zero
  write_file(fn, out);
}
// The subset writers (LDAUtils.java:1037-1122): the selection at the head of a file named and MAPPED (8*rows*cols bytes,
// the double writer's size, for the int variants too) for the whole matrix; the tail stays zero.  m is row-major [..][ld].
inline void pad_to(std::string &out, int64_t bytes) { if ((int64_t)out.size() < bytes) out.append((size_t)(bytes - (int64_t)out.size()), '\0'); }
inline void put_double_be(std::string &out, double v) { uint64_t bits; std::memcpy(&bits, &v, 8); put_be(out, bits, 8); }
inline std::string write_binary_double_matrix_rows(const double *m, int64_t ld, int iteration, int64_t rows, int64_t cols, const std::string &prefix,
                                                   const std::vector<int32_t> &row_indices) {                    // LDAUtils.java:1037-1051
  std::string out;
  for (int32_t r : row_indices) for (int64_t c = 0; c < cols; ++c) put_double_be(out, m[(int64_t)r * ld + c]);
  pad_to(out, 8 * rows * cols);
  const std::string fn = binary_matrix_name(prefix, rows, cols, iteration);
  write_file(fn, out);
  return fn;
}
inline std::string write_binary_int_matrix_rows(const int32_t *m, int64_t ld, int iteration, int64_t rows, int64_t cols, const std::string &prefix,
                                                const std::vector<int32_t> &row_indices) {                       // LDAUtils.java:1053-1067
  std::string out;
  for (int32_t r : row_indices) for (int64_t c = 0; c < cols; ++c) put_be(out, (uint32_t)m[(int64_t)r * ld + c], 4);
  pad_to(out, 8 * rows * cols);
  const std::string fn = binary_matrix_name(prefix, rows, cols, iteration);
  write_file(fn, out);
  return fn;
}
inline std::string write_binary_double_matrix_cols(const double *m, int64_t ld, int iteration, int64_t rows, int64_t cols, const std::string &prefix,
                                                   const std::vector<int32_t> &col_indices) {                    // LDAUtils.java:1069-1083
  std::string out;
  for (int64_t r = 0; r < rows; ++r) for (int32_t c : col_indices) put_double_be(out, m[r * ld + c]);
  pad_to(out, 8 * rows * cols);
  const std::string fn = binary_matrix_name(prefix, rows, cols, iteration);
  write_file(fn, out);
  return fn;
}
inline std::string write_binary_int_matrix_cols(const int32_t *m, int64_t ld, int iteration, int64_t rows, int64_t cols, const std::string &prefix,
                                                const std::vector<int32_t> &col_indices) {                       // LDAUtils.java:1108-1122
  std::string out;
  for (int64_t r = 0; r < rows; ++r) for (int32_t c : col_indices) put_be(out, (uint32_t)m[r * ld + c], 4);
  pad_to(out, 8 * rows * cols);
  const std::string fn = binary_matrix_name(prefix, rows, cols, iteration);
  write_file(fn, out);
  return fn;
}
// row r contributes m[r][indices[r][j]] for every j: the driver's Selected_Phi_KxV of each topic's top words (UPLDA:888)
inline std::string write_binary_double_matrix_indices(const double *m, int64_t ld, int iteration, const std::string &prefix,
                                                      const std::vector<std::vector<int32_t>> &indices) {        // LDAUtils.java:1085-1106
  const int64_t rows = (int64_t)indices.size(), cols = rows ? (int64_t)indices[0].size() : 0;
  std::string out;
  for (int64_t r = 0; r < rows; ++r) for (int32_t c : indices[(size_t)r]) put_double_be(out, m[r * ld + c]);
  pad_to(out, 8 * rows * cols);
  const std::string fn = binary_matrix_name(prefix, rows, cols, iteration);
  write_file(fn, out);
  return fn;
}
inline std::vector<int32_t> read_binary_int_matrix(int64_t rows, int64_t cols, const std::string &fn) {        // LDAUtils.java:1255-1267
  std::ifstream f(fn, std::ios::binary);
  std::vector<int32_t> m((size_t)rows * cols);
  for (auto &v : m) {
    unsigned char b[4];
    if (!f.read(reinterpret_cast<char *>(b), 4)) throw std::runtime_error("short file " + fn);
    v = (int32_t)(((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]);
  }
  return m;
}
inline std::vector<double> read_binary_double_matrix(int64_t rows, int64_t cols, const std::string &fn) {
  std::ifstream f(fn, std::ios::binary);
  std::vector<double> m((size_t)rows * cols);
  for (auto &v : m) {
    unsigned char b[8];
    if (!f.read(reinterpret_cast<char *>(b), 8)) throw std::runtime_error("short file " + fn);
    uint64_t bits = 0;
    for (int i = 0; i < 8; ++i) bits = (bits << 8) | b[i];
    std::memcpy(&v, &bits, 8);
  }
  return m;
}

// ---- integer text ----------------------------------------------------------------------------------------------
inline void write_ascii_int_matrix(const int32_t *m, int64_t rows, int64_t cols, const std::string &fn, const std::string &sep = ",") {   // LDAUtils.java:1175-1197
  std::string out;
  for (int64_t r = 0; r < rows; ++r) {
    for (int64_t c = 0; c < cols; ++c) {
      if (c) out += sep;
      out += std::to_string(m[r * cols + c]);
    }
    out += "\n";
  }
  write_file(fn, out);
}
// UPLDA:945-968 logTopicIndicators: z_<iteration>.csv, one document per line, an empty line for an empty document
inline std::string write_topic_indicators(const int64_t *doc_ptr, int64_t num_docs, const int32_t *z, const std::string &log_dir, int iteration) {
  const std::string fn = log_dir + "/z_" + std::to_string(iteration) + ".csv";
  std::string out;
  for (int64_t d = 0; d < num_docs; ++d) {
    for (int64_t i = doc_ptr[d]; i < doc_ptr[d + 1]; ++i) {
      if (i > doc_ptr[d]) out += ",";
      out += std::to_string(z[i]);
    }
    out += "\n";
  }
  write_file(fn, out);
  return fn;
}

// ---- doubles as text (parity unpinned, see the header) --------------------------------------------------------------
// |d| = 0.DIGITS x 10^e10 with the shortest digits that round-trip (what std::to_chars prints); a one-digit result is
// widened to the two-digit decimal closest to d, as Java does (Double.MIN_VALUE prints as 4.9E-324)
inline void shortest_digits(double d, std::string &digits, int &e10) {
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof buf, std::fabs(d), std::chars_format::scientific);
  std::string s(buf, r.ptr);                          // d[.ddd]e[+-]xx
  const size_t epos = s.find('e');
  const int ex = std::stoi(s.substr(epos + 1));
  digits.clear();
  for (size_t i = 0; i < epos; ++i)
    if (s[i] != '.') digits.push_back(s[i]);
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  e10 = ex + 1;
  if (digits.size() == 1) {
    char two[64];
    std::snprintf(two, sizeof two, "%.1e", std::fabs(d));   // correctly rounded (half-even on the exact value) two-digit decimal
    if (std::strtod(two, nullptr) == std::fabs(d)) {
      std::string t(two);
      const size_t ep = t.find('e');
      std::string dg;
      for (size_t i = 0; i < ep; ++i)
        if (t[i] != '.') dg.push_back(t[i]);
      while (dg.size() > 1 && dg.back() == '0') dg.pop_back();
      digits = dg;
      e10 = std::stoi(t.substr(ep + 1)) + 1;
    }
  }
}
inline std::string java_double_to_string(double d) {                       // java.lang.Double.toString
  if (d != d) return "NaN";
  if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
  if (d == 0) return std::signbit(d) ? "-0.0" : "0.0";
  std::string s;
  int e10;
  shortest_digits(d, s, e10);
  std::string body;
  const double a = std::fabs(d);
  if (a >= 1e-3 && a < 1e7) {
    if (e10 <= 0) body = "0." + std::string((size_t)(-e10), '0') + s;
    else if ((size_t)e10 >= s.size()) body = s + std::string((size_t)e10 - s.size(), '0') + ".0";
    else body = s.substr(0, (size_t)e10) + "." + s.substr((size_t)e10);
  } else {
    body = s.substr(0, 1) + "." + (s.size() > 1 ? s.substr(1) : std::string("0")) + "E" + std::to_string(e10 - 1);
  }
  return (std::signbit(d) ? "-" : "") + body;
}
// String.format("%.<places>f", d): HALF_UP on the shortest decimal digits
inline std::string java_format_fixed(double d, int places) {
  if (d != d) return "NaN";
  if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
  std::string s;
  int e10 = 0;
  if (d == 0) { s = "0"; e10 = 1; }
  else shortest_digits(d, s, e10);
  // value = 0.s x 10^e10; keep digits up to 10^-places
  std::string ip, fp;                                  // integer part, fraction digits (unbounded)
  if (e10 <= 0) { ip = "0"; fp = std::string((size_t)(-e10), '0') + s; }
  else if ((size_t)e10 >= s.size()) { ip = s + std::string((size_t)e10 - s.size(), '0'); fp = ""; }
  else { ip = s.substr(0, (size_t)e10); fp = s.substr((size_t)e10); }
  bool up = fp.size() > (size_t)places && fp[(size_t)places] >= '5';       // HALF_UP: the first dropped digit decides
  fp.resize((size_t)places, '0');
  std::string all = ip + fp;
  if (up) {
    int i = (int)all.size() - 1;
    while (i >= 0 && all[(size_t)i] == '9') all[(size_t)i--] = '0';
    if (i >= 0) all[(size_t)i]++;
    else all.insert(all.begin(), '1');
  }
  const size_t ilen = all.size() - (size_t)places;
  std::string out = all.substr(0, ilen);
  if (places > 0) out += "." + all.substr(ilen);
  return (std::signbit(d) ? "-" : "") + out;           // Java keeps the sign of a negative value that rounds to zero
}
// new DecimalFormat("00.###E0").format(d), d finite and non-zero: two integer digits, at most three fraction digits,
// HALF_EVEN on the exact binary value (glibc's printf rounds the exact value to nearest-even), exponent without a plus
inline std::string java_decimal_format_00_3e0(double d) {
  char buf[64];
  std::snprintf(buf, sizeof buf, "%.4e", std::fabs(d));                    // D.DDDDe[+-]XX: five significant digits
  std::string s(buf);
  const size_t epos = s.find('e');
  const int ex = std::stoi(s.substr(epos + 1));
  std::string dg;
  for (size_t i = 0; i < epos; ++i)
    if (s[i] != '.') dg.push_back(s[i]);
  std::string frac = dg.substr(2);
  while (!frac.empty() && frac.back() == '0') frac.pop_back();
  std::string out = dg.substr(0, 2);
  if (!frac.empty()) out += "." + frac;
  return (std::signbit(d) ? "-" : "") + out + "E" + std::to_string(ex - 1);
}
inline std::string format_double(double d, int places = 4) {                // LDAUtils.formatDouble, LDAUtils.java:1199-1207
  if ((0 < d && d < 0.0001) || (-0.0001 < d && d < 0)) return java_decimal_format_00_3e0(d);
  return java_format_fixed(d, places);
}
inline void write_ascii_double_matrix(const double *m, int64_t rows, int64_t cols, const std::string &fn, const std::string &sep = ",",
                                      int places = 4) {                    // LDAUtils.java:1222-1249 (Phi_KxV_*.csv, Theta_DxK_*.csv)
  std::string out;
  for (int64_t r = 0; r < rows; ++r) {
    for (int64_t c = 0; c < cols; ++c) {
      if (c) out += sep;
      out += format_double(m[r * cols + c], places);
    }
    out += "\n";
  }
  write_file(fn, out);
}
inline std::string ascii_matrix_name(const std::string &dir, const std::string &kind, int64_t rows, int64_t cols, int iteration) {   // UPLDA:757-762,805-811
  char tail[64];
  std::snprintf(tail, sizeof tail, "_%lld_%lld_%05d.csv", (long long)rows, (long long)cols, iteration);
  return dir + "/" + kind + tail;
}
inline void append_line(const std::string &fn, const std::string &line) {
  std::ofstream f(fn, std::ios::binary | std::ios::app);
  if (!f) throw std::runtime_error("cannot open " + fn);
  f << line << "\n";
}
inline void append_log_likelihood(const std::string &log_dir, int iteration, double log_lik) {          // LDAUtils.java:971-979
  append_line(log_dir + "/log-likelihood.txt", std::to_string(iteration) + "\t" + java_double_to_string(log_lik));
}
inline void append_heldout_log_likelihood(const std::string &log_dir, int iteration, double value) {   // LDAUtils.java:928-940
  append_line(log_dir + "/test_held_out_log_likelihood.txt", std::to_string(iteration) + "\t" + java_double_to_string(value));
}
inline void append_log_posterior(const std::string &log_dir, int iteration, double value, long long millis) {   // LDAUtils.java:955-968
  append_line(log_dir + "/log-posterior.txt", std::to_string(iteration) + "\t" + java_format_fixed(value, 6) + "\t" + std::to_string(millis));
}

}  // namespace formats
}  // namespace ggs
