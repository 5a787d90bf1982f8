"""`dasac::fast_div` (csrc/common.hpp): the multiply-and-shift that replaces the index divisions of the streaming kernels must equal
n / d for every 0 <= n < 2^31 the kernels can produce.  Host-side check (hipcc builds a small host program, no GPU): all divisors
1..4096, powers of two and their neighbours, random divisors up to 2^31 - 1, each against boundary and random numerators."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SRC = r"""
#include "common.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
static int host_fdiv(int n, dasac::FastDiv f) {      // what dasac::fdiv does on the device (__umulhi)
  return f.mul ? (int)((unsigned)(((unsigned long long)(unsigned)n * f.mul) >> 32) >> f.shift) : n;
}
int main() {
  std::vector<int> ds;
  for (int d = 1; d <= 4096; ++d) ds.push_back(d);
  for (int k = 12; k < 31; ++k) { ds.push_back((1 << k) - 1); ds.push_back(1 << k); ds.push_back((1 << k) + 1); }
  ds.push_back(2147483647);
  srand(7);
  for (int i = 0; i < 4000; ++i) ds.push_back(1 + (int)(((unsigned)rand() * 65536u + (unsigned)rand()) % 2147483646u));
  long long bad = 0, checked = 0;
  for (int d : ds) {
    const dasac::FastDiv f = dasac::fast_div(d);
    std::vector<long long> ns = {0, 1, d - 1ll, d, d + 1ll, 2ll * d - 1, 2ll * d, 2147483647ll, 2147483646ll,
                                 2147483647ll / d * d, 2147483647ll / d * d - 1};
    for (int i = 0; i < 64; ++i) ns.push_back(((long long)rand() * 65536 + rand()) % 2147483648ll);
    for (long long k : {3ll, 7ll, 1000ll, 65535ll}) ns.push_back(k * d - 1);
    for (long long n : ns) {
      if (n < 0 || n > 2147483647ll) continue;
      ++checked;
      if (host_fdiv((int)n, f) != (int)(n / d)) { if (bad++ < 5) printf("d=%d n=%lld got %d want %lld\n", d, n, host_fdiv((int)n, f), n / d); }
    }
  }
  printf("checked %lld bad %lld\n", checked, bad);
  return bad ? 1 : 0;
}
"""


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_fast_div_equals_integer_division(tmp_path):
    src = tmp_path / "fastdiv_check.hip"
    src.write_text(SRC)
    exe = tmp_path / "fastdiv_check"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "da-sac_amd", "csrc"), str(src), "-o", str(exe)], stderr=subprocess.DEVNULL)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-500:]
    assert "bad 0" in out.stdout
