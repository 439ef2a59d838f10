"""Short kernel-family names shared by scripts/trace_summary.py and scripts/trace_sequence.py."""
import re


def family(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if "at::native" in n or "rocclr" in n or "rocprim" in n or "at::cuda" in n:
        m = re.findall(r"(\w+Functor\w*|\w+_kernel_cuda|reduce_kernel|multi_tensor_apply|fill\w*|copy\w*|CatArray\w*|"
                       r"index\w*_kernel|distribution\w*|flip\w*)", n)
        return "aten:" + (m[0] if m else n[:40])
    return n.split("(")[0][:60]
