"""`python -m manta scene.py [args...]` -- stands in for the reference executable `manta scene.py args`
(pwrapper/pymain.cpp:39-131): sets sys.argv / args / SCENEFILE and runs the scene as __main__."""
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        print("Usage: python -m manta <scene.py> [args...]")
        return 1
    scene = sys.argv[1]
    sys.argv = sys.argv[1:]
    import mantaflow_amd.api as api
    api.args = sys.argv[1:]
    api.SCENEFILE = scene
    import manta
    manta.args, manta.SCENEFILE = api.args, scene
    runpy.run_path(scene, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
