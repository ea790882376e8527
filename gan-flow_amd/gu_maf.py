"""README.md:41,57 of the reference starts FFJORD as `python3 -m ffjord.gu_maf --gu_num {1,8}`; model_plot.py:2 imports
the same driver as `ffjord.gu_ffjord`.  Both names resolve to one implementation here."""
from .gu_ffjord import *  # noqa: F401,F403
from .gu_ffjord import main

if __name__ == "__main__":
    main()
