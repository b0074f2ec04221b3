"""Entrypoint with the reference's file name (reference tav_nn.py): `python tav_nn.py --preset B --batch_size 4 --epoch 1`."""
import tav_amd  # noqa: F401
from tav_amd.tav_nn import main, prepare_dataloader, runModel  # noqa: F401

if __name__ == "__main__":
    main()
