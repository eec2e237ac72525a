"""NRX_TEST_POISON=1: fill blocks of torch's caching allocator with 0xFF bytes (float NaN, int -1) and release them, so that the next torch.empty
buffers start from poison -- see tests/conftest.py::_poison_recycled_gpu_memory.  The stress checkers call poison() once per random launch."""
import os

ON = os.environ.get("NRX_TEST_POISON") == "1"


def poison(device="cuda:0"):
    if not ON:
        return
    import torch
    big = [torch.full((32 << 20,), 0xFF, dtype=torch.uint8, device=device) for _ in range(4)]
    small = [torch.full((256 << 10,), 0xFF, dtype=torch.uint8, device=device) for _ in range(64)]
    del big, small
