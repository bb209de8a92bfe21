"""`python -m idelucs` == `python -m idelucs_amd` (the reference CLI surface, idelucs/__main__.py:274-318)."""
from idelucs_amd.__main__ import main

if __name__ == "__main__":
    main()
