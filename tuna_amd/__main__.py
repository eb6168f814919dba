"""python -m tuna_amd "SPE : N N 1.0977 : HF CC-PVTZ : EXTREME"   (the reference: python3 TUNA/tuna.py ...)"""
import sys

from .energy import run
from ._lib import TunaError

if __name__ == "__main__":
    try:
        run(" ".join(sys.argv[1:]), silent=False)
    except TunaError as e:
        print(f"\n ERROR: {e}  :(\n")
        sys.exit(1)
