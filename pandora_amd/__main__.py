"""python -m pandora_amd config.json output_dir [-v]  (the reference's `pandora` command, Main.py)"""
import argparse

from . import main

if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Pandora stereo matching on MI355X (pandora_amd)")
    parser.add_argument("config", help="path to a json file with the input files paths and the pipeline parameters")
    parser.add_argument("output_dir", help="path to the output directory")
    parser.add_argument("-v", "--verbose", action="store_true")
    args = parser.parse_args()
    main(args.config, args.output_dir, args.verbose)
