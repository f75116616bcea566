#!/usr/bin/env python
"""source/test_habitat_s1.py of the reference: generation with the stage-1 model (see _gen_main.py)."""
from _gen_main import main_habitat

if __name__ == '__main__':
    main_habitat('s1')
