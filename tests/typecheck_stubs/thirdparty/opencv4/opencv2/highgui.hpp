// TYPE-CHECK STAND-IN, not OpenCV (see tests/typecheck_stubs/README.md): intentionally empty.
