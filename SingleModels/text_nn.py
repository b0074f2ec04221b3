"""Entrypoint with the reference's path (reference SingleModels/text_nn.py): `python SingleModels/text_nn.py --preset B --batch_size 16`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tav_amd  # noqa: E402,F401
from tav_amd.SingleModels.text_nn import main, prepare_dataloader, runModel  # noqa: E402,F401

if __name__ == "__main__":
    main()
