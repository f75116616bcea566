#!/usr/bin/env python
"""source/test_proxe_s2.py of the reference: generation with the stage-2 model (see _gen_main.py)."""
from _gen_main import main_proxe

if __name__ == '__main__':
    main_proxe('s2')
