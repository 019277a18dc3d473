/* test driver of LineChunkReader (wga_host.cpp): reads a file in pieces of `target` bytes, recycles every piece's buffer as the
 * command line's PAF pipeline does, and prints the number of pieces, bytes, lines and the position-weighted byte sum
 * sum((i + 1) * byte[i]) mod 2^64 of what it was handed —
 * the test compares them with Python's reading of the same file.  Not part of the product. */
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include "../wgatools_amd/host/wga_host.hpp"
int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string path = argv[1];
  const size_t target = (size_t)strtoull(argv[2], nullptr, 10);
  wga::LineChunkReader rd;
  rd.open(&path);
  std::string piece, prev;
  uint64_t h = 0, bytes = 0, pieces = 0;
  bool ends_ok = true;
  while (rd.next(piece, target)) {
    pieces++;
    for (unsigned char c : piece) h += ++bytes * (uint64_t)c;
    if (rd.bytes_before + piece.size() != rd.next_bytes) ends_ok = false;
    rd.recycle(std::move(prev)); /* the piece before this one is done with */
    prev = std::move(piece);
    piece = std::string();
  }
  printf("%llu %llu %llu %llu %d\n", (unsigned long long)pieces, (unsigned long long)bytes, (unsigned long long)rd.next_lines,
         (unsigned long long)h, ends_ok ? 1 : 0);
  return 0;
}
