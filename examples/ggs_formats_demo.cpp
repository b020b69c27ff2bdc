// ggs_formats_demo -- exercises include/ggs_formats.hpp for tests/test_formats.py:
//   text  < hex-bits-per-line      prints  Double.toString \t %.4f \t %.6f \t LDAUtils.formatDouble   for every value
//   files <dir>                    writes a small int matrix and a small double matrix (binary + ascii) and z_7.csv
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

#include "ggs_formats.hpp"

int main(int argc, char **argv) {
  namespace F = ggs::formats;
  if (argc >= 2 && std::string(argv[1]) == "text") {
    std::string line;
    while (std::getline(std::cin, line)) {
      if (line.empty()) continue;
      const uint64_t bits = std::stoull(line, nullptr, 16);
      double d;
      std::memcpy(&d, &bits, 8);
      std::printf("%s\t%s\t%s\t%s\n", F::java_double_to_string(d).c_str(), F::java_format_fixed(d, 4).c_str(), F::java_format_fixed(d, 6).c_str(),
                  d == 0 || d != d || std::isinf(d) ? F::java_format_fixed(d, 4).c_str() : F::format_double(d).c_str());
    }
    return 0;
  }
  if (argc >= 3 && std::string(argv[1]) == "files") {
    const std::string dir = argv[2];
    const int32_t im[6] = {1, -2, 300000, 0, 2147483647, -2147483647 - 1};
    const double dm[6] = {0.25, -1.5e-7, 3.14159265358979, 0.0, 1e300, 4.9e-324};
    F::write_binary_int_matrix(im, 2, 3, dir + "/" + F::binary_matrix_name("N", 2, 3, 12));
    F::write_binary_double_matrix(dm, 3, 2, dir + "/" + F::binary_matrix_name("phi", 3, 2, 12));
    F::write_ascii_int_matrix(im, 2, 3, dir + "/ints.csv");
    F::write_ascii_double_matrix(dm, 3, 2, F::ascii_matrix_name(dir, "Phi_KxV", 3, 2, 12));
    const int64_t doc_ptr[4] = {0, 2, 2, 5};
    const int32_t z[5] = {4, 0, 1, 1, 9};
    F::write_topic_indicators(doc_ptr, 3, z, dir, 7);
    F::append_log_likelihood(dir, 3, -123456.789);
    F::append_heldout_log_likelihood(dir, 3, -1.0e-5);
    F::append_log_posterior(dir, 3, -98765.4321987, 1700000000000LL);
    const double big[9] = {1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0};                      // the matrices of LDAUtilsTest.java:37-216
    const int32_t ibig[9] = {1, 2, 3, 4, 5, 6, 7, 8, 9};
    F::write_binary_double_matrix_rows(big, 3, 1, 3, 3, dir + "/drows", {0, 2});
    F::write_binary_int_matrix_rows(ibig, 3, 1, 3, 3, dir + "/irows", {0, 2});
    F::write_binary_double_matrix_cols(big, 3, 1, 3, 3, dir + "/dcols", {0, 2});
    F::write_binary_int_matrix_cols(ibig, 3, 1, 3, 3, dir + "/icols", {2, 1});
    F::write_binary_double_matrix_indices(big, 3, 1, dir + "/dsel", {{0, 2}, {1, 2}, {0, 1}});
    const auto back = F::read_binary_int_matrix(2, 3, dir + "/" + F::binary_matrix_name("N", 2, 3, 12));
    const auto dback = F::read_binary_double_matrix(3, 2, dir + "/" + F::binary_matrix_name("phi", 3, 2, 12));
    return (std::memcmp(back.data(), im, sizeof im) == 0 && std::memcmp(dback.data(), dm, sizeof dm) == 0) ? 0 : 3;
  }
  std::fprintf(stderr, "usage: ggs_formats_demo text | files <dir>\n");
  return 2;
}
