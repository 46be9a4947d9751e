"""SURVEY §5 (sanitizers on the host restatement): the whole CPU oracle runs under AddressSanitizer +
UndefinedBehaviorSanitizer (+ LeakSanitizer) on small seeded inputs that reach the border paths (features next to
every image edge, empty sets, keypoint lists that overflow, NaN remap coordinates). CPU only — GPU sanitizers are
not available on this pool. The parity statements rest on this C code: memory errors or UB in it would make
"GPU == oracle" meaningless."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "sanitize"])
    exe = os.path.join(ROOT, "oracle", "_san", "sanitize_main")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "oracle sanitize run ok" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
