"""Benchmark suite: latency / throughput / online serving (reference: vllm/benchmarks/*, benchmarks/*.py)."""
