"""The CPU model tests once more with every engine on the GPU (fp32, PyTorch reference ops): their token-exact
assertions against transformers / written-out arithmetic then cover what the CPU runs cannot see -- tensors that towers,
projectors and processors build on the wrong device.  (The HIP kernels have their own tests: test_ops_gpu.py,
test_engine_gpu.py.)  Runs as a child process because the device switch is read when the test session starts
(tests/conftest.py, VDA_TEST_DEVICE)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["tests/test_multimodal_cpu.py", "tests/test_renamed_decoders_cpu.py", "tests/test_phi3_small_cpu.py",
         "tests/test_plamo2_cpu.py", "tests/test_dual_chunk_cpu.py"]


@pytest.mark.gpu
def test_cpu_model_tests_pass_with_engines_on_the_gpu():
    env = dict(os.environ, VDA_TEST_DEVICE="cuda", VDA_FORCE_TORCH_OPS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", *FILES, "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider",
                        "--timeout", "180", "--timeout-method", "thread"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-40:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


@pytest.mark.gpu
def test_quantised_checkpoint_tests_pass_with_engines_on_the_gpu():
    """Pre-quantised bitsandbytes / NVFP4 / microscaling-fp4 checkpoints and in-flight NF4 with the engine on the GPU in
    fp32: loading, sharding into the fused projections, the chunk layout the GPU keeps for code-book layers and its
    expansion must still give transformers' tokens exactly."""
    env = dict(os.environ, VDA_TEST_DEVICE="cuda", VDA_FORCE_TORCH_OPS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_quant_cpu.py", "-k",
                        "bitsandbytes or nvfp4 or mxfp4", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider",
                        "--timeout", "180", "--timeout-method", "thread"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-40:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
