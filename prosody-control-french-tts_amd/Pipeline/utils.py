"""Text extraction from TextGrid content (``Code/Pipeline/utils.py:5-28``): every ``text = "..."`` line
contributes its value with ``[annotations]``, commas and semicolons removed; empty and single-space
values are skipped; the pieces are joined by one space."""
import re

_ANNOT = re.compile(r"\[.*?\]")


def extract_clean_text_from_textgrid(textgrid_content: str) -> str:
    out = []
    for line in textgrid_content.split("\n"):
        if "text = " not in line:
            continue
        value = line.split("=")[1].strip().strip('"')        # (the reference keeps only what lies between the first two '=')
        if value and value != " ":
            out.append(_ANNOT.sub("", value).replace(",", "").replace(";", ""))
    return " ".join(out)
