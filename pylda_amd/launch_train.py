"""`python -m pylda_amd.launch_train ...`: the reference's training command line
(see pylda_amd/cli.py) on the MI355X engine."""
import sys

from pylda_amd.cli import train_main as main

if __name__ == "__main__":
    sys.exit(main())
