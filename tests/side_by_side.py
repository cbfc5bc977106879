#!/usr/bin/env python
"""The benchmark scripts of examples/ on the CPU restatement (oracle/), for side-by-side numbers next to the engine's.

examples/ is product-adjacent and knows the HIP engine only; the restatement is test infrastructure and may be
imported from tests/ alone, so the switch lives here:

    python tests/side_by_side.py random_miqp [--repeat 10] [--out results/random_miqp_cpu.csv]
    python tests/side_by_side.py power_converter [steps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

from oracle import oracle  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "random_miqp"
    if what == "random_miqp":
        import random_miqp
        random_miqp.main(sys.argv[2:], backend=oracle)
    elif what == "power_converter":
        import power_converter
        row, pc = power_converter.run(oracle, steps=int(sys.argv[2]) if len(sys.argv) > 2 else None)
        print({k: row[k] for k in sorted(row)})
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
