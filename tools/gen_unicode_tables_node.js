#!/usr/bin/env node
// Generates csrc/unicode_data.inc — the Unicode data the DEVICE compiler's regex front-end reads — from node's ICU (VERDICT r5 #5):
// an implementation of the Unicode Character Database that shares nothing with perl's Unicode::UCD, which oracle/unicode_data.inc comes
// from (tools/gen_unicode_tables.pl + tools/merge_unicode_delta.py). node v12.22.9 carries ICU 70.1 = Unicode 14.0; perl 5.34 = 13.0; the
// reference's regex-syntax 0.8.8 (Cargo.lock:1717-1720) a newer one still (DESIGN.md D19).
//
// DATA, not code — the same tables and the same file format as the perl generator writes:
//   general categories, scripts, Alphabetic / White_Space / Lowercase / Uppercase / Join_Control: every scalar value asked through
//     /\p{..}/u (one pass over a string that holds all 0x10F800 scalar values in order: a match is a RANGE);
//   simple case folding orbits: /c/iu (ECMAScript's Canonicalize under the u flag = CaseFolding.txt statuses C + S) for every code
//     point that has a case mapping of any kind, again against the string of all scalar values.
// The NAMES of the tables (aliases a pattern may use) are identifiers of the standard, listed below; the membership is what is generated.
//
// usage: node tools/gen_unicode_tables_node.js > pingoo_amd/csrc/unicode_data.inc
'use strict';

const MAX = 0x10FFFF;
// all scalar values in one string; offset (UTF-16 units) <-> code point
let parts = [];
for (let base = 0; base <= MAX; base += 0x1000) {
  let s = '';
  for (let cp = base; cp < base + 0x1000; cp++) if (cp < 0xD800 || cp > 0xDFFF) s += String.fromCodePoint(cp);
  parts.push(s);
}
const ALL = parts.join('');
parts = null;
function cpAt(off) {  // code point whose first UTF-16 unit sits at `off`
  if (off < 0xD800) return off;
  if (off < 0x10000 - 0x800) return off + 0x800;
  return 0x10000 + ((off - (0x10000 - 0x800)) >> 1);
}
function endCp(off) {  // last code point of a match that ends (exclusive) at `off`
  return cpAt(off) - 1 === 0xDFFF ? 0xD7FF : cpAt(off) - 1;
}
function rangesOf(re) {  // re: /(?:..)+/gu over ALL -> [[lo, hi], ...] over scalar values (a run across the surrogate gap is split)
  const out = [];
  let m;
  re.lastIndex = 0;
  while ((m = re.exec(ALL)) !== null) {
    const lo = cpAt(m.index), hi = m.index + m[0].length >= ALL.length ? MAX : endCp(m.index + m[0].length);
    if (lo <= 0xD7FF && hi >= 0xE000) { out.push([lo, 0xD7FF]); out.push([0xE000, hi]); }
    else out.push([lo, hi]);
  }
  return out;
}

// General_Category values: long name, short name (PropertyValueAliases.txt). LC = Lu | Ll | Lt; the one-letter groups are unions.
const GC = [['Cased_Letter', 'LC'], ['Close_Punctuation', 'Pe'], ['Connector_Punctuation', 'Pc'], ['Control', 'Cc', 'cntrl'], ['Currency_Symbol', 'Sc'], ['Dash_Punctuation', 'Pd'],
  ['Decimal_Number', 'Nd', 'digit'], ['Enclosing_Mark', 'Me'], ['Final_Punctuation', 'Pf'], ['Format', 'Cf'], ['Initial_Punctuation', 'Pi'], ['Letter', 'L'], ['Letter_Number', 'Nl'],
  ['Line_Separator', 'Zl'], ['Lowercase_Letter', 'Ll'], ['Mark', 'M', 'Combining_Mark'], ['Math_Symbol', 'Sm'], ['Modifier_Letter', 'Lm'], ['Modifier_Symbol', 'Sk'], ['Nonspacing_Mark', 'Mn'],
  ['Number', 'N'], ['Open_Punctuation', 'Ps'], ['Other', 'C'], ['Other_Letter', 'Lo'], ['Other_Number', 'No'], ['Other_Punctuation', 'Po'], ['Other_Symbol', 'So'], ['Paragraph_Separator', 'Zp'],
  ['Private_Use', 'Co'], ['Punctuation', 'P', 'punct'], ['Separator', 'Z'], ['Space_Separator', 'Zs'], ['Spacing_Mark', 'Mc'], ['Surrogate', 'Cs'], ['Symbol', 'S'], ['Titlecase_Letter', 'Lt'],
  ['Unassigned', 'Cn'], ['Uppercase_Letter', 'Lu']];
// Script values of Unicode 14.0: long name, ISO 15924 code (+ the two historical aliases Qaac / Qaai)
const SC = 'Adlam Adlm|Ahom Ahom|Anatolian_Hieroglyphs Hluw|Arabic Arab|Armenian Armn|Avestan Avst|Balinese Bali|Bamum Bamu|Bassa_Vah Bass|Batak Batk|Bengali Beng|Bhaiksuki Bhks|Bopomofo Bopo|Brahmi Brah|Braille Brai|Buginese Bugi|Buhid Buhd|Canadian_Aboriginal Cans|Carian Cari|Caucasian_Albanian Aghb|Chakma Cakm|Cham Cham|Cherokee Cher|Chorasmian Chrs|Common Zyyy|Coptic Copt Qaac|Cuneiform Xsux|Cypriot Cprt|Cypro_Minoan Cpmn|Cyrillic Cyrl|Deseret Dsrt|Devanagari Deva|Dives_Akuru Diak|Dogra Dogr|Duployan Dupl|Egyptian_Hieroglyphs Egyp|Elbasan Elba|Elymaic Elym|Ethiopic Ethi|Georgian Geor|Glagolitic Glag|Gothic Goth|Grantha Gran|Greek Grek|Gujarati Gujr|Gunjala_Gondi Gong|Gurmukhi Guru|Han Hani|Hangul Hang|Hanifi_Rohingya Rohg|Hanunoo Hano|Hatran Hatr|Hebrew Hebr|Hiragana Hira|Imperial_Aramaic Armi|Inherited Zinh Qaai|Inscriptional_Pahlavi Phli|Inscriptional_Parthian Prti|Javanese Java|Kaithi Kthi|Kannada Knda|Katakana Kana|Kayah_Li Kali|Kharoshthi Khar|Khitan_Small_Script Kits|Khmer Khmr|Khojki Khoj|Khudawadi Sind|Lao Laoo|Latin Latn|Lepcha Lepc|Limbu Limb|Linear_A Lina|Linear_B Linb|Lisu Lisu|Lycian Lyci|Lydian Lydi|Mahajani Mahj|Makasar Maka|Malayalam Mlym|Mandaic Mand|Manichaean Mani|Marchen Marc|Masaram_Gondi Gonm|Medefaidrin Medf|Meetei_Mayek Mtei|Mende_Kikakui Mend|Meroitic_Cursive Merc|Meroitic_Hieroglyphs Mero|Miao Plrd|Modi Modi|Mongolian Mong|Mro Mroo|Multani Mult|Myanmar Mymr|Nabataean Nbat|Nandinagari Nand|New_Tai_Lue Talu|Newa Newa|Nko Nkoo|Nushu Nshu|Nyiakeng_Puachue_Hmong Hmnp|Ogham Ogam|Ol_Chiki Olck|Old_Hungarian Hung|Old_Italic Ital|Old_North_Arabian Narb|Old_Permic Perm|Old_Persian Xpeo|Old_Sogdian Sogo|Old_South_Arabian Sarb|Old_Turkic Orkh|Old_Uyghur Ougr|Oriya Orya|Osage Osge|Osmanya Osma|Pahawh_Hmong Hmng|Palmyrene Palm|Pau_Cin_Hau Pauc|Phags_Pa Phag|Phoenician Phnx|Psalter_Pahlavi Phlp|Rejang Rjng|Runic Runr|Samaritan Samr|Saurashtra Saur|Sharada Shrd|Shavian Shaw|Siddham Sidd|SignWriting Sgnw|Sinhala Sinh|Sogdian Sogd|Sora_Sompeng Sora|Soyombo Soyo|Sundanese Sund|Syloti_Nagri Sylo|Syriac Syrc|Tagalog Tglg|Tagbanwa Tagb|Tai_Le Tale|Tai_Tham Lana|Tai_Viet Tavt|Takri Takr|Tamil Taml|Tangsa Tnsa|Tangut Tang|Telugu Telu|Thaana Thaa|Thai Thai|Tibetan Tibt|Tifinagh Tfng|Tirhuta Tirh|Toto Toto|Ugaritic Ugar|Unknown Zzzz|Vai Vaii|Vithkuqi Vith|Wancho Wcho|Warang_Citi Wara|Yezidi Yezi|Yi Yiii|Zanabazar_Square Zanb'
  .split('|').map((x) => x.split(' '));
const BIN = ['Alphabetic', 'White_Space', 'Lowercase', 'Uppercase', 'Join_Control'];

const norm = (n) => n.toLowerCase().replace(/[_\- ]/g, '');
const tables = [];  // [kind, names[], ranges]
for (const names of GC) {
  const r = names[1] === 'Cs' ? [] : rangesOf(new RegExp('\\p{gc=' + names[1] + '}+', 'gu'));
  tables.push([0, names, r]);
}
for (const names of SC) tables.push([1, names, rangesOf(new RegExp('\\p{sc=' + names[0] + '}+', 'gu'))]);
for (const p of BIN) tables.push([2, [p], rangesOf(new RegExp('\\p{' + p + '}+', 'gu'))]);

const out = [];
out.push("// GENERATED by tools/gen_unicode_tables_node.js from node's ICU (ICU " + process.versions.icu + ', Unicode ' + process.versions.unicode + '). Data, not code.');
out.push("// kind: 0 = General_Category value, 1 = Script value, 2 = binary property. Names are matched loosely (case, '_', '-', ' ' ignored).");
const flat = [], index = [];
for (const [kind, names, r] of tables) {
  const first = flat.length;
  for (const x of r) flat.push(x);
  const seen = [];
  for (const n of names.map(norm)) if (!seen.includes(n)) seen.push(n);
  index.push('{' + kind + ', "' + seen.join('|') + '", ' + first + ', ' + r.length + '},');
}
out.push('static const unsigned kUniRanges[][2] = {');
let line = '';
flat.forEach((x, i) => {
  line += '{0x' + x[0].toString(16).toUpperCase() + ',0x' + x[1].toString(16).toUpperCase() + '},';
  if (i % 8 === 7) { out.push(line); line = ''; }
});
out.push(line);
out.push('};');
out.push('struct UniTable { int kind; const char *names; unsigned first, count; };');
out.push('static const UniTable kUniTables[] = {');
for (const x of index) out.push(x);
out.push('};');

// simple case folding orbits: every code point with a case mapping of any kind, matched case-insensitively against all scalar values
const cased = rangesOf(/[\p{Cased}\p{Changes_When_Casefolded}\p{Changes_When_Casemapped}\p{Changes_When_Lowercased}\p{Changes_When_Uppercased}\p{Changes_When_Titlecased}]+/gu);
const esc = (cp) => '\\u{' + cp.toString(16) + '}';
const pairs = [];
for (const [lo, hi] of cased) {
  for (let cp = lo; cp <= hi; cp++) {
    const re = new RegExp(esc(cp), 'giu');
    let m;
    while ((m = re.exec(ALL)) !== null) {
      const other = cpAt(m.index);
      if (other !== cp) pairs.push([cp, other]);
    }
  }
}
pairs.sort((a, b) => a[0] - b[0] || a[1] - b[1]);
out.push('// (cp, other member of its simple case folding orbit), sorted by cp');
out.push('static const unsigned kUniFold[][2] = {');
line = '';
pairs.forEach((x, i) => {
  line += '{0x' + x[0].toString(16).toUpperCase() + ',0x' + x[1].toString(16).toUpperCase() + '},';
  if (i % 8 === 7) { out.push(line); line = ''; }
});
out.push(line);
out.push('};');
process.stdout.write(out.join('\n') + '\n');
