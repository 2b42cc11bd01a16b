"""Fingerprint of the kernel sources (rajni-vit_amd/csrc/* and include/*.h): what ties a committed PMC profile to the
code it was taken on.  `bench.py` drops `roofline.traffic` / `mfma_busy_frac` when the newest profile's fingerprint is
not the working tree's; tools/pmc_*.py store it in their JSON."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_fingerprint(root=ROOT):
    h = hashlib.sha256()
    for d in (os.path.join(root, "rajni-vit_amd", "csrc"), os.path.join(root, "include")):
        for fn in sorted(os.listdir(d)):
            if fn.endswith((".hip", ".h")):
                h.update(fn.encode())
                with open(os.path.join(d, fn), "rb") as f:
                    h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_fingerprint())
