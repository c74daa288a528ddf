from .entrypoints.cli.main import main

main()
