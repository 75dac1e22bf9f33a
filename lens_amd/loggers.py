"""Run folder, log file and device selection (behaviour of lens/src/loggers.py:7-65).

Creates ./lens/output/<ddmmyy-HH-MM-SS>/lens.log, echoes to the console, prints the banner and
returns the torch device: the ROCm GPU when one is present.  Unlike the reference there is no CPU
execution path behind the returned device; LENS.__init__ refuses to continue without a GPU.
"""
import logging
import os
from datetime import datetime

import torch

_BANNER = ["", "LENS on MI355X (lens_amd): Locational Encoding with Neuromorphic Systems",
           "hot path of AdamDHines/LENS v0.1.1 re-implemented as HIP kernels for gfx950", ""]


def model_logger(model):
    base = "./lens/output/"
    model.output_folder = base + datetime.now().strftime("%d%m%y-%H-%M-%S")
    os.makedirs(base, exist_ok=True)
    # The reference's folder name has 1-second granularity and a second run within the same second
    # dies in os.mkdir (loggers.py:13,19); on a GPU two runs per second are normal, so a suffix is added.
    stem, n = model.output_folder, 1
    while True:
        try:
            os.mkdir(model.output_folder)
            break
        except FileExistsError:
            n += 1
            model.output_folder = f"{stem}_{n}"
    model.logger = logging.getLogger("LENS")
    if model.logger.hasHandlers():
        model.logger.handlers.clear()
    model.logger.setLevel(logging.DEBUG)
    fh = logging.FileHandler(os.path.join(model.output_folder, "lens.log"), mode="a+")
    fh.setFormatter(logging.Formatter("%(asctime)-15s %(levelname)-8s %(message)s"))
    model.logger.addHandler(fh)
    model.logger.addHandler(logging.StreamHandler())
    for line in _BANNER:
        model.logger.info(line)
    if torch.cuda.is_available():
        name = torch.cuda.get_device_name(torch.cuda.current_device())
        model.logger.info(f"Current device is {name}")
        device = torch.device("cuda")
    else:
        model.logger.info("Current device is CPU")
        device = torch.device("cpu")
    model.logger.info("")
    return device
