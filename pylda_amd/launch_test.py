"""`python -m pylda_amd.launch_test ...`: the reference's held-out evaluation
command line (see pylda_amd/cli.py) on the MI355X engine."""
import sys

from pylda_amd.cli import test_main as main

if __name__ == "__main__":
    sys.exit(main())
