"""Text extraction from TextGrid content (``Code/Pipeline/utils.py:5-28``): every ``text = "..."`` line
contributes its value with ``[annotations]``, commas and semicolons removed; empty and single-space
values are skipped; the pieces are joined by one space."""
import glob
import re
from pathlib import Path

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


def save_clean_transcriptions_from_textgrids(input_dir, output_dir) -> None:
    """``Code/Pipeline/utils.py:30-55``: one ``<stem>.txt`` per ``*.TextGrid`` (any letter case) holding its cleaned text; a
    file that cannot be read is reported and skipped."""
    input_dir, output_dir = Path(input_dir), Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    for textgrid_path in glob.glob(str(input_dir / "*.[Tt][Ee][Xx][Tt][Gg][Rr][Ii][Dd]")):
        try:
            content = Path(textgrid_path).read_text(encoding="utf-8")
            (output_dir / (Path(textgrid_path).stem + ".txt")).write_text(extract_clean_text_from_textgrid(content), encoding="utf-8")
        except Exception as e:                                               # noqa: BLE001
            print(f"Error processing {textgrid_path}: {e}")
