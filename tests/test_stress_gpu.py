"""GPU tier: the random-program stress tools (tools/stress_lanes.py, tools/stress_ranks.py), a few seeded seconds each (VERDICT r3
#9 / SURVEY.md 5, race detection: the host side of the library -- lanes x enqueue threads x pairing x twins x automatic redraws,
and the multi-rank feeders / tickets -- had no stress in the tiers).  Every synchronous frame of a run is compared with what a
fresh single-lane context draws; a data race shows up as a mismatch, a crash, a hang (the timeout) or an error return."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, seconds=5, env=None, timeout=240):
    e = dict(os.environ)
    e.update({"STRESS_SECONDS": str(seconds)})
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + [str(a) for a in args], capture_output=True, text=True,
                       timeout=timeout, env=e, cwd=ROOT)
    tail = (p.stdout.strip().splitlines() or [""])[-1]
    assert p.returncode == 0 and tail.startswith("stress ok"), (p.stdout[-2000:], p.stderr[-3000:])
    print(script, args, tail)
    return tail


@pytest.mark.parametrize("seed", [41, 42])
def test_random_program_on_one_context_lanes_pairs_threads(seed):
    _run("stress_lanes.py", seed)


def test_random_program_with_near_only_sorts():
    _run("stress_lanes.py", 43, env={"STRESS_NEAR": "1"})


@pytest.mark.skipif(os.environ.get("GS_SKIP_SLOW") == "1", reason="large configs")
def test_random_program_on_four_million_splats_with_the_speculative_stash():
    """4 M splats along an orbit with jumps: near-only sorts through the chunk stashes and through the depth pass' own candidate stash
    (round 4), their misses, redraws and the back-off, mixed with synchronous frames that are checked against a fresh context"""
    tail = _run("stress_lanes.py", 44, seconds=12, env={"STRESS_BIG": "1"}, timeout=400)
    assert "sorts from the depth pass' own stash" in tail


@pytest.mark.parametrize("world,seed", [(2, 51), (3, 52)])
def test_random_program_over_ranks_in_process_transport(world, seed):
    _run("stress_ranks.py", seed, world)


# ---- the same programs with the library's HOST code under ThreadSanitizer (tools/build_tsan.sh: -fsanitize=thread on the host pass of
# hipcc, the gfx950 code objects unchanged; clang's TSan runtime preloaded into the interpreter; the uninstrumented HIP / ROCr runtimes
# suppressed, tests/tsan.supp).  Round 4's first run of this found a real one: a lane's enqueue thread read its error state outside the
# mutex while lane_drain() of the twin lane reset it.
def _tsan_lib():
    import glob
    lib = os.path.join(ROOT, "aframe-gaussian-splatting_amd", "csrc", "libgs_variant_tsan.so")
    srcs = glob.glob(os.path.join(ROOT, "aframe-gaussian-splatting_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "aframe-gaussian-splatting_amd", "csrc", "*.h")) + \
        glob.glob(os.path.join(ROOT, "aframe-gaussian-splatting_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "include", "gs_splat.h")]
    # (round 6: this pool's GPU boxes refuse sanitizer builds -- the script and its output are listed in .gpurunignore, so on a box they
    # are absent and the test is skipped; where they are present -- a workstation with its own GPU -- it runs as in rounds 4-5)
    if not os.path.exists(os.path.join(ROOT, "tools", "build_tsan.sh")):
        pytest.skip("tools/build_tsan.sh is not shipped to this box (the GPU pool refuses sanitizer builds)")
    # the runtime first: without one the build cannot link, and the test is to be skipped, not to fail on the link error
    rt = subprocess.run([os.path.join(ROOT, "tools", "build_tsan.sh"), "--runtime"], capture_output=True, text=True).stdout.strip()
    if not rt or not os.path.exists(rt):
        pytest.skip("no ThreadSanitizer runtime next to hipcc")
    if not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in srcs):
        subprocess.run([os.path.join(ROOT, "tools", "build_tsan.sh")], check=True, capture_output=True, timeout=900)
    return lib, rt


@pytest.mark.parametrize("script,args,extra", [("stress_lanes.py", (41,), {}), ("stress_ranks.py", (52, 3), {}),
                                               ("stress_lanes.py", (45,), {"STRESS_BIG": "1", "STRESS_SECONDS": "10"})])
def test_host_threading_under_thread_sanitizer(script, args, extra):
    if extra and os.environ.get("GS_SKIP_SLOW") == "1":
        pytest.skip("large configs")
    lib, rt = _tsan_lib()
    if not rt or not os.path.exists(rt):
        pytest.skip("no ThreadSanitizer runtime next to hipcc")
    e = dict(os.environ)
    e.update({"STRESS_SECONDS": "4"})
    e.update(extra)
    e.update({"GS_SPLAT_LIB": lib, "LD_PRELOAD": rt,
              "TSAN_OPTIONS": "report_signal_unsafe=0 exitcode=66 history_size=4 suppressions=" + os.path.join(ROOT, "tests", "tsan.supp")})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + [str(a) for a in args], capture_output=True, text=True, timeout=400, env=e, cwd=ROOT)
    out = p.stdout + p.stderr
    races = [l for l in out.splitlines() if l.startswith("SUMMARY: ThreadSanitizer")]
    assert p.returncode == 0 and not races and "stress ok" in p.stdout, (races[:10], out[-3000:])
    print(script, args, (p.stdout.strip().splitlines() or [""])[-1])
